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

// compact variant: row r of x is token live[r] (the dropout counter stays the token's own element index,
// so the mask is the one the dense layout would draw)
__global__ __launch_bounds__(256) void gather_dropout_compact_kernel(unsigned d4, const int64_t* ids, const int* live,
                                                                     const int* n_live, const float* table, Dropout drop,
                                                                     float* x) {
    // 32-bit index arithmetic (the launcher checks M * d/4 < 2^32): a 64-bit division per element costs as
    // much as the Philox call
    const unsigned total4 = (unsigned)(*n_live) * d4;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const unsigned r = i / d4, c4 = i - r * d4;
        const long t = live[r];
        f32x4 v = *reinterpret_cast<const f32x4*>(table + ids[t] * (long)(4 * d4) + 4 * c4);
        if (drop.thresh != 0u) v *= dropout_scale4(drop.seed, 0u, (uint64_t)(t * d4 + c4), drop.thresh, drop.inv_keep);
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

int launch_gather_dropout_compact(long M, int d, const int64_t* ids, const int* live, const int* n_live,
                                  const float* table, const Dropout& drop, float* x, hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    if (M * (d / 4) >= (1L << 32)) { set_error("gather: %ld float4 elements overflow the 32-bit index", M * (d / 4)); return NRMS_EINVAL; }
    int blocks = cdiv(M * (d / 4), 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("gather_dropout", stream);
    hipLaunchKernelGGL(gather_dropout_compact_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned)(d / 4), ids, live,
                       n_live, table, drop, x);
    return check_launch("gather_dropout");
}

// qkv[m, :] = bias for padding tokens: with an all-zero embedding row 0 their x row is exactly zero
// (also under dropout), so the projection of a padding token IS the bias and the GEMM skips them.
constexpr int FILL_SLICES = 6;          // 64 lanes x 6 float4 = 1536 floats >= 3 d_model (d_model <= 512)
template <bool SKIP_ALL_PAD>
__global__ __launch_bounds__(256) void fill_pad_rows_kernel_t(long n_seq, int S, int n4, const int64_t* ids, const float* row,
                                                              float* out) {
    // one wave per sequence: its padding rows get the bias; a sequence that is ALL padding is skipped (with
    // skip_all_pad: the attention kernels take the closed form for it and never read its rows)
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
    f32x4 b[FILL_SLICES];
#pragma unroll
    for (int i = 0; i < FILL_SLICES; ++i)
        b[i] = (lane + 64 * i < n4) ? reinterpret_cast<const f32x4*>(row)[lane + 64 * i] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (long q = wave; q < n_seq; q += n_waves) {
        const bool is_pad = lane < S && ids[q * S + lane] == 0;
        unsigned long long m = __ballot(is_pad);
        if (SKIP_ALL_PAD && __popcll(m) == S) continue;
        while (m != 0ull) {
            const int r = __builtin_ctzll(m);
            m &= m - 1ull;
            f32x4* o = reinterpret_cast<f32x4*>(out) + (q * S + r) * n4;
#pragma unroll
            for (int i = 0; i < FILL_SLICES; ++i)
                if (lane + 64 * i < n4) o[lane + 64 * i] = b[i];
        }
    }
}

// skip_all_pad must match what the attention kernels do (they skip all-padding sequences only without a mask)
int launch_fill_pad_rows(long n_seq, int S, int n, const int64_t* ids, const float* row, float* out, bool skip_all_pad,
                         hipStream_t stream) {
    if (n_seq <= 0) return NRMS_OK;
    if ((n & 3) != 0 || n > 256 * FILL_SLICES || S > 64) { set_error("fill_pad_rows: n=%d S=%d unsupported", n, S); return NRMS_EINVAL; }
    int blocks = cdiv(n_seq, 4);
    if (blocks > 256 * 32) blocks = 256 * 32;
    TimingScope ts("fill_pad_rows", stream);
    if (skip_all_pad) hipLaunchKernelGGL(fill_pad_rows_kernel_t<true>, dim3(blocks), dim3(256), 0, stream, n_seq, S, n / 4, ids, row, out);
    else hipLaunchKernelGGL(fill_pad_rows_kernel_t<false>, dim3(blocks), dim3(256), 0, stream, n_seq, S, n / 4, ids, row, out);
    return check_launch("fill_pad_rows");
}

// Token positions whose id is not the padding id, in ASCENDING order (deterministic: the order fixes the
// summation order of the weight-gradient GEMM that runs over the compact rows), plus the inverse map.
// Three small passes: per-block counts, one-block exclusive scan, ordered write.
//   live[0 .. *n_live) = positions m with ids[m] != 0;  pos[m] = index of m in live, or -1 (may be null)
constexpr int CP_BLOCK = 1024;         // tokens per block
__global__ __launch_bounds__(256) void compact_count_kernel(long M, const int64_t* ids, int* counts) {
    __shared__ int wsum[4];
    const long m0 = (long)blockIdx.x * CP_BLOCK;
    int c = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long m = m0 + p * 256 + threadIdx.x;
        c += (m < M && ids[m] != 0) ? 1 : 0;
    }
    c = (int)wave_sum((float)c);        // <= 256 per wave: exact in fp32
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void compact_scan_kernel(int n_blocks, int* counts, int* n_live) {
    // exclusive scan of counts[0 .. n_blocks) in place (one block, serial over chunks of 1024)
    __shared__ int buf[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_blocks ? counts[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) counts[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_live = carry;
}

__global__ __launch_bounds__(256) void compact_write_kernel(long M, const int64_t* ids, const int* offsets, int* live,
                                                            int* pos) {
    __shared__ int wave_cnt[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m0 = (long)blockIdx.x * CP_BLOCK;
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
    }
    __syncthreads();
    const int base = offsets[blockIdx.x];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long m = m0 + p * 256 + threadIdx.x;
        const int r = base + wave_cnt[p][wave] + pre[p];
        if (flag[p]) live[r] = (int)m;
        if (pos != nullptr && m < M) pos[m] = flag[p] ? r : -1;
    }
}

size_t compact_scratch_ints(long M) { return (size_t)cdiv(M, CP_BLOCK) + 64; }

int launch_compact_live_rows(long M, const int64_t* ids, int* live, int* pos, int* n_live, int* scratch,
                             hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    const int nb = cdiv(M, CP_BLOCK);
    TimingScope ts("compact_rows", stream);
    hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(256), 0, stream, M, ids, scratch);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream, nb, scratch, n_live);
    hipLaunchKernelGGL(compact_write_kernel, dim3(nb), dim3(256), 0, stream, M, ids, scratch, live, pos);
    return check_launch("compact_live_rows");
}

// dtable[ids[t], :] += dx[r, :] * keep(t, :) / (1 - p) for the compact rows r < *n_live, t = live[r].
template <typename IDX>
__global__ __launch_bounds__(256) void scatter_dropout_compact_kernel(unsigned d, const int64_t* ids, const int* live,
                                                                      const int* n_live, const float* dx, Dropout drop,
                                                                      float* dtable) {
    // IDX = unsigned where M * d < 2^32 (a 64-bit division per element costs as much as the Philox call)
    const IDX total = (IDX)(*n_live) * d;
    const IDX stride = (IDX)gridDim.x * blockDim.x;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const IDX r = i / d;
        const unsigned c = (unsigned)(i - r * d);
        const long t = live[r];
        const long id = ids[t];
        float v = dx[i];
        if (drop.thresh != 0u) v *= dropout_scale1(drop.seed, 0u, (uint64_t)(t * d + c), drop.thresh, drop.inv_keep);
        atomicAdd(dtable + id * (long)d + c, v);
    }
}

// dtable[ids[t], :] += dx[t, :] * keep(t, :) / (1 - p) for the live tokens t = live[r], r < *n_live (dx has one row per
// TOKEN): the fp16 mode's dense path (embedding row 0 not zero).  Float atomics: order-dependent in the last bits.
__global__ __launch_bounds__(256) void scatter_dense_rows_kernel(unsigned d, const int64_t* ids, const int* live,
                                                                 const int* n_live, const float* dx, Dropout drop, float* dtable) {
    const unsigned long total = (unsigned long)(*n_live) * d;
    const unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned long r = i / d;
        const unsigned c = (unsigned)(i - r * d);
        const long t = live[r];
        float v = dx[t * (long)d + c];
        if (drop.thresh != 0u) v *= dropout_scale1(drop.seed, 0u, (uint64_t)(t * d + c), drop.thresh, drop.inv_keep);
        atomicAdd(dtable + ids[t] * (long)d + c, v);
    }
}

int launch_scatter_dense_rows(long M, int d, const int64_t* ids, const int* live, const int* n_live, const float* dx,
                              const Dropout& drop, float* dtable, hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    int blocks = cdiv(M * d, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("scatter_dropout", stream);
    hipLaunchKernelGGL(scatter_dense_rows_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned)d, ids, live, n_live, dx, drop, dtable);
    return check_launch("scatter_dense_rows");
}

int launch_scatter_dropout_compact(long M, int d, const int64_t* ids, const int* live, const int* n_live, const float* dx,
                                   const Dropout& drop, float* dtable, hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    int blocks = cdiv(M * d, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("scatter_dropout", stream);
    if (M * d < (1L << 32))
        hipLaunchKernelGGL(scatter_dropout_compact_kernel<unsigned>, dim3(blocks), dim3(256), 0, stream, (unsigned)d, ids,
                           live, n_live, dx, drop, dtable);
    else
        hipLaunchKernelGGL(scatter_dropout_compact_kernel<unsigned long>, dim3(blocks), dim3(256), 0, stream, (unsigned)d, ids,
                           live, n_live, dx, drop, dtable);
    return check_launch("scatter_dropout");
}

// ---------------------------------------------------------------------------------------
// Grouped scatter: the live tokens are bucketed by word id (histogram -> exclusive scan -> placement), then one
// wave per vocabulary row sums the dX rows of its bucket and adds the result to the table gradient with plain
// stores.  Against the atomic scatter: no float atomics at all (the 88 M of them were its bound), one Philox
// call per float4 instead of per float, and every table-gradient row is written exactly once.
__global__ __launch_bounds__(256) void tok_hist_kernel(const int64_t* ids, const int* live, const int* n_live, int* cnt) {
    const int n = *n_live;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
        atomicAdd(cnt + ids[live[r]], 1);
}

// exclusive scan of n ints in place, two small kernels: block-local scans of 1024 elements + block totals, then
// every block adds the sum of the totals before it (n / 1024 <= a few dozen blocks; a single-block version took 73 us)
__global__ __launch_bounds__(1024) void scan_local_kernel(int n, int* data, int* bsum) {
    __shared__ int buf[1024];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int v = i < n ? data[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    if (i < n) data[i] = buf[threadIdx.x] - v;
    if (threadIdx.x == 1023) bsum[blockIdx.x] = buf[1023];
}

__global__ __launch_bounds__(1024) void scan_add_kernel(int n, int n_blocks, int* data, const int* bsum, int* total) {
    __shared__ int base;
    if (threadIdx.x == 0) {
        int s = 0;
        for (int b = 0; b < (int)blockIdx.x; ++b) s += bsum[b];
        base = s;
        if (blockIdx.x == n_blocks - 1) *total = s + bsum[blockIdx.x];
    }
    __syncthreads();
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) data[i] += base;
}

__global__ __launch_bounds__(256) void tok_place_kernel(const int64_t* ids, const int* live, const int* n_live,
                                                        const int* offs, int* cursor, int* order) {
    const int n = *n_live;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const long id = ids[live[r]];
        order[offs[id] + atomicAdd(cursor + id, 1)] = r;
    }
}

// H16: dX rows are fp16 [rows][ldx] still multiplied by the fp16 backward's loss scale (half the bytes of the dX GEMM's
// store and of this kernel's reads); the sum is taken in fp32 and divided by the scale (sc[1], device) once per table row.
typedef _Float16 sg_h4 __attribute__((ext_vector_type(4)));
template <bool H16>
__global__ __launch_bounds__(256) void scatter_grouped_kernel(int V, int d4, const int* live, const int* offs,
                                                              const int* total, const int* order, const void* dxv, int ldx,
                                                              const float* sc, Dropout drop, float* dtable) {
    const float* dx = reinterpret_cast<const float*>(dxv);
    const _Float16* dx16 = reinterpret_cast<const _Float16*>(dxv);
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V || v == 0) return;                                   // id 0 = padding: no gradient
    const int beg = offs[v], end = v + 1 < V ? offs[v + 1] : *total;
    if (beg == end) return;
    // a lane owns float4 columns lane, lane + 64, ... (SL slices): ONE pass over the bucket, all slices of a
    // token's row loaded together (a pass per slice doubled the dependent order -> row load chain)
    constexpr int SL = 2;                          // d_model <= 512 -> d4 <= 128
    f32x4 s[SL];
#pragma unroll
    for (int j = 0; j < SL; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto add_row = [&](int r) {
        const long t = live[r];
#pragma unroll
        for (int j = 0; j < SL; ++j) {
            const int c4 = lane + 64 * j;
            if (c4 < d4) {
                f32x4 g;
                if (H16) {
                    const sg_h4 gh = *reinterpret_cast<const sg_h4*>(dx16 + (long)r * ldx + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = (float)gh[e];
                } else {
                    g = *reinterpret_cast<const f32x4*>(dx + ((long)r * d4 + c4) * 4);
                }
                if (drop.thresh != 0u) g *= dropout_scale4(drop.seed, 0u, (uint64_t)(t * d4 + c4), drop.thresh, drop.inv_keep);
                s[j] += g;
            }
        }
    };
    // The placement fills a bucket in whatever order its atomics resolve.  Each 64-entry chunk of the bucket is
    // therefore consumed in ASCENDING row order (wave-wide minimum selection): a word that occurs at most 64
    // times in the step -- all of them in practice outside a handful of stop words -- gets a bit-reproducible
    // gradient row; longer buckets are reproducible up to the order of their chunks.
    for (int k0 = beg; k0 < end; k0 += 64) {
        const int n = min(64, end - k0);
        int mine = lane < n ? order[k0 + lane] : 0x7fffffff;
        for (int i = 0; i < n; ++i) {
            int m = mine;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
            add_row(m);
            if (mine == m) mine = 0x7fffffff;
        }
    }
#pragma unroll
    for (int j = 0; j < SL; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < d4) *reinterpret_cast<f32x4*>(dtable + ((long)v * d4 + c4) * 4) += H16 ? s[j] * sc[1] : s[j];
    }
}

size_t scatter_grouped_scratch_ints(long M, int V) { return (size_t)2 * (V + 64) + (size_t)M + 64; }

// The lists the grouped scatter walks (per table row: the compact rows of its occurrences, ascending) depend on the ids only:
// a caller may build them early, on another stream (the fp16 backward does, beside its fused kernels), and pass prepared = true.
int launch_scatter_prepare(long M, int V, const int64_t* ids, const int* live, const int* n_live, int* scratch, hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    int* cnt = scratch;                       // [V] counts -> exclusive offsets (in place)
    int* cursor = cnt + V + 64;               // [V]
    int* total = cursor + V;                  // [1]
    int* order = cursor + V + 64;             // [M]
    if (hipMemsetAsync(cnt, 0, (size_t)(2 * (V + 64)) * sizeof(int), stream) != hipSuccess) {
        set_error("scatter_grouped: memset failed");
        return NRMS_ELAUNCH;
    }
    int blocks = cdiv(M, 256);
    if (blocks > 2048) blocks = 2048;
    TimingScope ts("scatter_prepare", stream);
    hipLaunchKernelGGL(tok_hist_kernel, dim3(blocks), dim3(256), 0, stream, ids, live, n_live, cnt);
    {
        const int nb = cdiv(V, 1024);
        int* bsum = order;                    // [nb] block totals: the bucket array is free until the placement
        hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(1024), 0, stream, V, cnt, bsum);
        hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(1024), 0, stream, V, nb, cnt, bsum, total);
    }
    hipLaunchKernelGGL(tok_place_kernel, dim3(blocks), dim3(256), 0, stream, ids, live, n_live, cnt, cursor, order);
    return check_launch("scatter_prepare");
}

// scratch: scatter_grouped_scratch_ints(M, V) ints.  dx is compact (row r belongs to token live[r]).
int launch_scatter_grouped(long M, int V, int d, const int64_t* ids, const int* live, const int* n_live, const void* dx,
                           const Dropout& drop, float* dtable, int* scratch, hipStream_t stream, bool dx_fp16, int ldx,
                           const float* sc, bool prepared) {
    if (M <= 0) return NRMS_OK;
    if (!prepared) {
        const int rc = launch_scatter_prepare(M, V, ids, live, n_live, scratch, stream);
        if (rc) return rc;
    }
    int* cnt = scratch;
    int* cursor = cnt + V + 64;
    int* total = cursor + V;
    int* order = cursor + V + 64;
    TimingScope ts("scatter_dropout", stream);
    if (dx_fp16)
        hipLaunchKernelGGL(scatter_grouped_kernel<true>, dim3(cdiv(V, 4)), dim3(256), 0, stream, V, d / 4, live, cnt, total, order, dx,
                           ldx, sc, drop, dtable);
    else
        hipLaunchKernelGGL(scatter_grouped_kernel<false>, dim3(cdiv(V, 4)), dim3(256), 0, stream, V, d / 4, live, cnt, total, order, dx,
                           d, nullptr, drop, dtable);
    return check_launch("scatter_grouped");
}

// dst[i] = src[i] if 0 <= src[i] < vocab else 0 (the padding id); *n_bad += number of ids replaced.  The encoder
// kernels index the table, the histogram and the placement arrays with the raw id: every id stream goes through
// here first (nn.Embedding raises on an out-of-range index, nrms_v0.py:134-139).
template <class SRC>
__global__ __launch_bounds__(256) void sanitize_ids_kernel(long n, const SRC* src, int64_t* dst, long vocab, int* n_bad) {
    int bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int64_t v = (int64_t)src[i];
        const bool ok = v >= 0 && v < vocab;
        bad += ok ? 0 : 1;
        dst[i] = ok ? v : 0;
    }
    const unsigned long long m = __ballot(bad != 0);
    if (m != 0ull) {                                   // rare path
        bad = (int)wave_sum((float)bad);               // <= 64 * few per wave: exact in fp32
        if ((threadIdx.x & 63) == 0) atomicAdd(n_bad, bad);
    }
}

// src: int64 ids (the reference's batch dict), or int32 ids (half the bytes over PCIe for a caller that keeps them so)
int launch_sanitize_ids(long n, const void* src, bool src_is_int32, int64_t* dst, int vocab, int* n_bad, hipStream_t stream) {
    if (n <= 0) return NRMS_OK;
    int blocks = cdiv(n, 256 * 4);
    if (blocks > 256 * 8) blocks = 256 * 8;
    TimingScope ts("sanitize_ids", stream);
    if (src_is_int32)
        hipLaunchKernelGGL(sanitize_ids_kernel<int32_t>, dim3(blocks), dim3(256), 0, stream, n, (const int32_t*)src, dst, (long)vocab, n_bad);
    else
        hipLaunchKernelGGL(sanitize_ids_kernel<int64_t>, dim3(blocks), dim3(256), 0, stream, n, (const int64_t*)src, dst, (long)vocab, n_bad);
    return check_launch("sanitize_ids");
}

// Distinct titles of a batch, exactly and without a sort: an open-addressing table of row indices.  One lane per
// title hashes its L word ids and probes; an empty slot is claimed with a CAS (the claimant becomes the group's
// representative and takes the next unique index), an occupied slot is compared word by word with the row that owns
// it (ids is read-only, so the owner's words are always valid), and a mismatch -- two different titles in one slot --
// moves on to the next slot.  Padding slots (the same all-zero title tens of thousands of times at C = 300) take the
// read-only path: the slot is loaded before any CAS is tried.
// The all-zero row is handled per wavefront: one lane probes for all of them (a hundred thousand lanes loading one
// slot with a device-scope load is a hot spot at the memory side: 2 ms instead of 0.05).
__global__ __launch_bounds__(256) void dedup_insert_kernel(long n, int L, const int64_t* ids, int* table, unsigned mask,
                                                           int* inverse, int* rep_rows, int* n_unique) {
    const int lane = threadIdx.x & 63;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int64_t* row = ids + t * L;
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)L;
        int64_t any = 0;
        for (int i = 0; i < L; ++i) {
            const int64_t w = row[i];
            any |= w;
            uint64_t k = (uint64_t)w * 0xFF51AFD7ED558CCDull;
            k ^= k >> 32;
            h = (h ^ k) * 0xC4CEB9FE1A85EC53ull;
            h ^= h >> 29;
        }
        h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33;
        const bool pad = any == 0;
        const unsigned long long pads = __ballot(pad);
        const int leader = pads != 0 ? __ffsll((long long)pads) - 1 : 0;
        int result = 0;
        if (!pad || lane == leader) {
            unsigned slot = (unsigned)h & mask;
            for (;;) {                                               // the table has >= 2n slots: an empty one is always reached
                int cur = __hip_atomic_load(table + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur < 0) {
                    cur = atomicCAS(table + slot, -1, (int)t);
                    if (cur < 0) {                                   // claimed: representative of a new group
                        const int u = atomicAdd(n_unique, 1);
                        rep_rows[u] = (int)t;
                        result = u;
                        break;
                    }
                }
                const int64_t* other = ids + (long)cur * L;
                bool same = true;
                for (int i = 0; i < L; ++i) same = same && other[i] == row[i];
                if (same) { result = -cur - 1; break; }              // resolved to the owner's unique index below
                slot = (slot + 1) & mask;
            }
        }
        const int lead = __shfl(result, leader);                     // the owner's index, or the pointer to the owner
        inverse[t] = (pad && lane != leader) ? lead : result;
    }
}

__global__ __launch_bounds__(256) void dedup_resolve_kernel(long n, int* inverse) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int v = inverse[t];
        if (v < 0) inverse[t] = inverse[-v - 1];                     // owners hold their (non-negative) index and are not rewritten
    }
}

int launch_title_dedup(long n, int L, const int64_t* ids, int* table, long table_size, int* inverse, int* rep_rows,
                       int* n_unique, hipStream_t stream) {
    if (hipMemsetAsync(n_unique, 0, sizeof(int), stream) != hipSuccess) { set_error("title_dedup: memset failed"); return NRMS_ELAUNCH; }
    if (n <= 0) return NRMS_OK;
    if (hipMemsetAsync(table, 0xFF, (size_t)table_size * sizeof(int), stream) != hipSuccess) {
        set_error("title_dedup: memset failed");
        return NRMS_ELAUNCH;
    }
    int blocks = cdiv(n, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("title_dedup", stream);
    hipLaunchKernelGGL(dedup_insert_kernel, dim3(blocks), dim3(256), 0, stream, n, L, ids, table, (unsigned)(table_size - 1),
                       inverse, rep_rows, n_unique);
    hipLaunchKernelGGL(dedup_resolve_kernel, dim3(blocks), dim3(256), 0, stream, n, inverse);
    return check_launch("title_dedup");
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

}  // namespace nrms
