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

// Token positions whose id is not the padding id.  Their order is irrelevant (rows of dX are
// independent and the scatter is a sum), so each 1024-token block just claims a range of the output
// with one atomic; inside a block positions stay ascending (wave ballots), which keeps neighbouring
// rows of the compact dX neighbouring in dQKV.
__global__ __launch_bounds__(256) void compact_live_rows_kernel(long M, const int64_t* ids, int* live, int* n_live) {
    __shared__ int wave_cnt[4][4];
    __shared__ int base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m0 = (long)blockIdx.x * 1024;
    bool flag[4];
    int pre[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long m = m0 + p * 256 + threadIdx.x;
        flag[p] = m < M && ids[m] != 0;
        const unsigned long long b = __ballot(flag[p]);
        pre[p] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[p][wave] = __popcll(b);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int p = 0; p < 4; ++p)
            for (int w = 0; w < 4; ++w) { const int c = wave_cnt[p][w]; wave_cnt[p][w] = tot; tot += c; }
        base = atomicAdd(n_live, tot);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p)
        if (flag[p]) live[base + wave_cnt[p][wave] + pre[p]] = (int)(m0 + p * 256 + threadIdx.x);
}

int launch_compact_live_rows(long M, const int64_t* ids, int* live, int* n_live, hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    if (hipMemsetAsync(n_live, 0, sizeof(int), stream) != hipSuccess) { set_error("compact_live_rows: memset failed"); return NRMS_ELAUNCH; }
    TimingScope ts("compact_rows", stream);
    hipLaunchKernelGGL(compact_live_rows_kernel, dim3(cdiv(M, 1024)), dim3(256), 0, stream, M, ids, live, n_live);
    return check_launch("compact_live_rows");
}

// dtable[ids[t], :] += dx[r, :] * keep(t, :) / (1 - p) for the compact rows r < *n_live, t = live[r].
__global__ __launch_bounds__(256) void scatter_dropout_compact_kernel(unsigned d, const int64_t* ids, const int* live,
                                                                      const int* n_live, const float* dx, Dropout drop,
                                                                      float* dtable) {
    const unsigned long total = (unsigned long)(*n_live) * d;
    const unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned long r = i / d;
        const unsigned c = (unsigned)(i - r * d);
        const long t = live[r];
        const long id = ids[t];
        float v = dx[i];
        if (drop.thresh != 0u) v *= dropout_scale1(drop.seed, 0u, (uint64_t)(t * d + c), drop.thresh, drop.inv_keep);
        atomicAdd(dtable + id * (long)d + c, v);
    }
}

int launch_scatter_dropout_compact(long M, int d, const int64_t* ids, const int* live, const int* n_live, const float* dx,
                                   const Dropout& drop, float* dtable, hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    int blocks = cdiv(M * d, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("scatter_dropout", stream);
    hipLaunchKernelGGL(scatter_dropout_compact_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned)d, ids, live, n_live,
                       dx, drop, dtable);
    return check_launch("scatter_dropout");
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
