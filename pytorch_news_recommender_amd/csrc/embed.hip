// Word-embedding gather (+dropout) and its gradient scatter-add: the two HBM/atomic-bound ends of
// the news encoder (nn.Embedding.from_pretrained(padding_idx=0) + nn.Dropout,
// /root/reference/MIND_2020/model/nrms_v0.py:134-139,166, and its dense backward, which the
// reference runs 55 times per step -- SURVEY.md a-1).
#include "gemm.h"

namespace nrms {

// x[m, :] = table[ids[m], :] * keep(m, :) / (1 - p).   One float4 per lane, rows walked in
// order: every wave-instruction reads/writes whole 16-byte chunks of at most two table rows.
__global__ __launch_bounds__(256) void gather_dropout_kernel(unsigned total4, unsigned d4, const int64_t* ids,
                                                             const float* table, Dropout drop, float* x) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const unsigned m = i / d4, c4 = i - m * d4;
        const long id = ids[m];
        f32x4 v = *reinterpret_cast<const f32x4*>(table + id * (long)(4 * d4) + 4 * c4);
        if (drop.thresh != 0u) v *= dropout_scale4(drop.seed, 0u, (uint64_t)i, drop.thresh, drop.inv_keep);
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

// dtable[ids[m], :] += dx[m, :] * keep(m, :) / (1 - p), rows with id 0 skipped (padding_idx).
// One float per lane so that a wave-instruction adds 256 contiguous bytes of one table row (the
// shape global_atomic_add_f32 runs at full rate on gfx950); float atomics are order-dependent in
// the last bits, as any parallel reduction of the reference's dense gradient would be.
__global__ __launch_bounds__(256) void scatter_dropout_kernel(unsigned long total, unsigned d, const int64_t* ids,
                                                              const float* dx, Dropout drop, float* dtable) {
    const unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned long m = i / d;
        const unsigned c = (unsigned)(i - m * d);
        const long id = ids[m];
        if (id == 0) continue;
        float v = dx[i];
        if (drop.thresh != 0u) v *= dropout_scale1(drop.seed, 0u, (uint64_t)i, drop.thresh, drop.inv_keep);
        atomicAdd(dtable + id * (long)d + c, v);
    }
}

int launch_gather_dropout(long M, int d, const int64_t* ids, const float* table, const Dropout& drop, float* x,
                          hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    const long total4 = M * (d / 4);
    if (total4 >= (1L << 32)) { set_error("gather: %ld float4 elements overflow the 32-bit index", total4); return NRMS_EINVAL; }
    int blocks = cdiv(total4, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("gather_dropout", stream);
    hipLaunchKernelGGL(gather_dropout_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned)total4, (unsigned)(d / 4),
                       ids, table, drop, x);
    return check_launch("gather_dropout");
}

int launch_scatter_dropout(long M, int d, const int64_t* ids, const float* dx, const Dropout& drop, float* dtable,
                           hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    const long total = M * d;
    int blocks = cdiv(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("scatter_dropout", stream);
    hipLaunchKernelGGL(scatter_dropout_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned long)total, (unsigned)d,
                       ids, dx, drop, dtable);
    return check_launch("scatter_dropout");
}

}  // namespace nrms
