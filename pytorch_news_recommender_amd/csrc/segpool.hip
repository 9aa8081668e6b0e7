// Gather + additive-attention aggregate over variable-length segments (include/nrms_hip.h: nrms_segment_pool_fwd / _bwd).
//
//   out[s] = sum_{k in segment s} alpha_k x[idx_k],   alpha = softmax_{k in s}( q_vec . tanh(W_add x[idx_k] + b_add) )
//
// SURVEY section 8 f-4 (BASELINE configs 4-5): the aggregation step of a HieRec-style hierarchical interest model (clicked news
// pooled per sub-topic, sub-topic interests per topic, topic interests per user: every level is this operation over a partition of
// its rows) and of a user-news graph encoder (neighbour gather + attention aggregate: the segments are adjacency lists, a row
// may be a member of many).  The reference holds no implementation of either (model/tanr.py is empty): the checker is a torch
// restatement of the formula above (oracle/segpool_oracle.py), PARITY UNPINNED.
//
// The attention logit of a member depends on its row only, so the additive projection runs ONCE over the rows -- the
// NRMS additive attention's own GEMMs (model/nrms_v0.py:100-126; split-bf16 or f32 MFMA, gemm*.hip), not once per segment --
// and what is per segment is HBM-bound index work: a wave per segment gathers logits and rows through the index list
// (coalesced 16-byte row reads, four members in flight), softmax by wave reductions, weighted row sum in registers.
//   forward : T = tanh(x W^T + b) [rows, q], logit = T q_vec [rows]      (NT GEMM + seg_logit)
//             alpha [nnz], out [n_seg, d]                                 (seg_pool_fwd)
//   backward: d(alpha_k) = <dout_s, x_k>, d(logit) by the softmax rule    (seg_da), d(q_vec) = sum_r d(logit_r) T_r (fixed-order
//             partial sums),
//             d(W_add), d(b_add) = dZ^T [x | 1] and dx = dZ W_add with dZ = d(logit) q_vec (1 - T^2)   (the TN / NT GEMMs' dZ loaders),
//             dx[idx_k] += alpha_k dout_s                                  (seg_scatter)
//   A row listed by several segments (a graph: no NRMS_SEGPOOL_ROWS_UNIQUE) collects its members' shares in a FIXED order, without
//   atomics: the list entries are sorted by row (stable radix sort, rocPRIM: the one library call of this file -- index
//   preparation, not arithmetic), and a workgroup per row adds its entries in ascending list position (seg_row_da, seg_row_gather).
#include <algorithm>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "gemm.h"

namespace nrms {

constexpr int SEG_WPB = 4;                      // waves (= segments / rows) per workgroup

// T[r][:] = tanh(T[r][:]) in place, logit[r] = <q_vec, T[r]>; one wave per row, q % 4 == 0, q <= 1024
__global__ __launch_bounds__(64 * SEG_WPB) void seg_logit_kernel(long n_rows, int q, float* T, const float* q_vec, float* logit) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * SEG_WPB + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    float* row = T + r * q;
    float acc = 0.f;
    for (int c = 4 * lane; c < q; c += 256) {
        f32x4 z = *reinterpret_cast<const f32x4*>(row + c);
        const f32x4 qq = *reinterpret_cast<const f32x4*>(q_vec + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { z[e] = fast_tanh(z[e]); acc += qq[e] * z[e]; }
        *reinterpret_cast<f32x4*>(row + c) = z;
    }
    acc = wave_sum(acc);
    if (lane == 0) logit[r] = acc;
}

// one wave per segment; d % 4 == 0, d <= 1024
__global__ __launch_bounds__(64 * SEG_WPB) void seg_pool_fwd_kernel(long n_seg, int d, const float* x, const int* seg_ptr, const int* idx,
                                                                    const float* logit, float* alpha, float* out) {
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * SEG_WPB + (threadIdx.x >> 6);
    if (s >= n_seg) return;
    const int p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
    const int d4 = d >> 2;
    float* orow = out + s * d;
    if (p1 <= p0) {                                 // an empty segment aggregates nothing
        for (int c = lane; c < d4; c += 64) *reinterpret_cast<f32x4*>(orow + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    float mx = -3.0e38f;
    for (int k = p0 + lane; k < p1; k += 64) mx = fmaxf(mx, logit[idx[k]]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = p0 + lane; k < p1; k += 64) {
        const float e = __expf(logit[idx[k]] - mx);
        alpha[k] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int k = p0 + lane; k < p1; k += 64) alpha[k] *= inv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // weighted row sum: lane owns float4 columns lane, lane + 64, ... ; four members' rows in flight
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = p0; k0 < p1; k0 += 4) {
        float a[4];
        const float* xr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u < p1 ? k0 + u : p1 - 1;
            a[u] = k0 + u < p1 ? alpha[k] : 0.f;
            xr[u] = x + (long)idx[k] * d;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < d4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xr[u] + 4 * c);
                    acc[j] += v * a[u];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < d4) *reinterpret_cast<f32x4*>(orow + 4 * c) = acc[j];
    }
}

// d(logit) of every member: da_k = alpha_k (<dout_s, x_k> - sum_j alpha_j <dout_s, x_j>).  UNIQUE (a row has one membership at
// most): stored to da_row[idx_k].  Otherwise: left per entry in dal[k] with seg_of[k] = s, for the per-row sums below.
template <bool UNIQUE>
__global__ __launch_bounds__(64 * SEG_WPB) void seg_da_kernel(long n_seg, int d, const float* x, const int* seg_ptr, const int* idx,
                                                              const float* alpha, const float* dout, float* dal, float* da_row, int* seg_of) {
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * SEG_WPB + (threadIdx.x >> 6);
    if (s >= n_seg) return;
    const int p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
    if (p1 <= p0) return;
    const int d4 = d >> 2;
    f32x4 dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        dv[j] = c < d4 ? *reinterpret_cast<const f32x4*>(dout + s * d + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float sumad = 0.f;                               // (the same value in every lane)
    for (int k = p0; k < p1; ++k) {
        const float* xr = x + (long)idx[k] * d;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < d4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c);
                dot += v[0] * dv[j][0] + v[1] * dv[j][1] + v[2] * dv[j][2] + v[3] * dv[j][3];
            }
        }
        dot = wave_sum(dot);
        if (lane == 0) dal[k] = dot;
        sumad += alpha[k] * dot;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k = p0 + lane; k < p1; k += 64) {
        const float da = alpha[k] * (dal[k] - sumad);
        if (UNIQUE) da_row[idx[k]] = da;
        else { dal[k] = da; seg_of[k] = (int)s; }
    }
}

// ---- rows listed by several segments: the entries of a row, in ascending list position (tval, bounded by tptr)
__global__ __launch_bounds__(256) void seg_tkey_kernel(long cap, const int* idx, const int* nnz_dev, int n_rows, int* key, int* val) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= cap) return;
    key[e] = e < *nnz_dev ? idx[e] : n_rows;         // (capacity beyond the lists' real length sorts to the end)
    val[e] = (int)e;
}

// tptr[r] = first position of the sorted keys that is >= r, r = 0 .. n_rows
__global__ __launch_bounds__(256) void seg_tptr_kernel(long cap, int n_rows, const int* key_sorted, int* tptr) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r > n_rows) return;
    long lo = 0, hi = cap;
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (key_sorted[mid] < r) lo = mid + 1; else hi = mid; }
    tptr[r] = (int)lo;
}

// da_row[r] = sum of its entries' d(logit), lanes striding the row's list, then the wave reduction: a fixed order
__global__ __launch_bounds__(64 * SEG_WPB) void seg_row_da_kernel(long n_rows, const int* tptr, const int* tval, const float* dal, float* da_row) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * SEG_WPB + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int p0 = tptr[r], p1 = tptr[r + 1];
    float acc = 0.f;
    for (int j = p0 + lane; j < p1; j += 64) acc += dal[tval[j]];
    acc = wave_sum(acc);
    if (lane == 0) da_row[r] = acc;
}

// dx[r][:] += sum over the row's entries e (ascending list position) of alpha_e dout[seg_of[e]][:], as a segmented reduction over
// the SORTED entries so that a row everybody lists (a popular news of a click graph holds 10 % of all entries) is not one
// wave's serial loop: a wave walks a span of 64 sorted positions; a row whose entries all lie inside the span is finished there
// (single writer); a row that crosses span boundaries leaves one partial sum per span (slot 0: its run starts at the span's
// first position, slot 1: later) and seg_row_combine adds them in span order.
constexpr int SEG_SPAN = 64;
__global__ __launch_bounds__(64 * SEG_WPB) void seg_span_gather_kernel(long n_spans, int n_rows, int d, const int* key_sorted, const int* tval,
                                                                       const int* tptr, const int* seg_of, const float* alpha, const float* dout,
                                                                       float* partial, float* dx) {
    const int lane = threadIdx.x & 63;
    const long span = (long)blockIdx.x * SEG_WPB + (threadIdx.x >> 6);
    if (span >= n_spans) return;
    const long pos = span * SEG_SPAN + lane;
    const int row_l = key_sorted[pos];                                  // (the arrays hold n_spans * 64 entries: the capacity is padded)
    const int e_l = tval[pos];
    const bool live_l = row_l < n_rows;
    const float a_l = live_l ? alpha[e_l] : 0.f;
    const int s_l = live_l ? seg_of[e_l] : 0;
    const int d4 = d >> 2;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int run_start = 0;
    for (int j = 0; j < SEG_SPAN; ++j) {
        const int row = __shfl(row_l, j, 64);
        if (row >= n_rows) break;                                       // sorted: the rest is padding
        const float a = __shfl(a_l, j, 64);
        const float* src = dout + (long)__shfl(s_l, j, 64) * d;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int c = lane + 64 * c4;
            if (c < d4) acc[c4] += *reinterpret_cast<const f32x4*>(src + 4 * c) * a;
        }
        const int next = j + 1 < SEG_SPAN ? __shfl(row_l, j + 1, 64) : -1;
        if (next != row) {                                              // the run [run_start, j] of `row` ends here
            const int p0 = tptr[row], p1 = tptr[row + 1];
            const bool owned = p0 / SEG_SPAN == (p1 - 1) / SEG_SPAN;
            float* o = owned ? dx + (long)row * d : partial + (span * 2 + (run_start == 0 ? 0 : 1)) * d;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int c = lane + 64 * c4;
                if (c < d4) {
                    f32x4 v = acc[c4];
                    if (owned) v += *reinterpret_cast<const f32x4*>(o + 4 * c);
                    *reinterpret_cast<f32x4*>(o + 4 * c) = v;
                    acc[c4] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            run_start = j + 1;
        }
    }
}

__global__ __launch_bounds__(64 * SEG_WPB) void seg_row_combine_kernel(long n_rows, int d, const int* tptr, const float* partial, float* dx) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * SEG_WPB + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int p0 = tptr[r], p1 = tptr[r + 1];
    if (p1 <= p0) return;
    const int sa = p0 / SEG_SPAN, sb = (p1 - 1) / SEG_SPAN;
    if (sa == sb) return;                                               // finished by its span
    const int d4 = d >> 2;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = sa; i <= sb; ++i) {
        const float* src = partial + ((long)i * 2 + ((i == sa && p0 != sa * SEG_SPAN) ? 1 : 0)) * d;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int c = lane + 64 * c4;
            if (c < d4) acc[c4] += *reinterpret_cast<const f32x4*>(src + 4 * c);
        }
    }
    float* o = dx + r * d;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        const int c = lane + 64 * c4;
        if (c < d4) *reinterpret_cast<f32x4*>(o + 4 * c) = *reinterpret_cast<const f32x4*>(o + 4 * c) + acc[c4];
    }
}

// dx[idx_k][:] += alpha_k dout_s[:]   (segments that partition the rows: a row is written by one wave)
__global__ __launch_bounds__(64 * SEG_WPB) void seg_scatter_kernel(long n_seg, int d, const int* seg_ptr, const int* idx, const float* alpha,
                                                                   const float* dout, float* dx) {
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * SEG_WPB + (threadIdx.x >> 6);
    if (s >= n_seg) return;
    const int p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
    const int d4 = d >> 2;
    f32x4 dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        dv[j] = c < d4 ? *reinterpret_cast<const f32x4*>(dout + s * d + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int k = p0; k < p1; ++k) {
        const float a = alpha[k];
        float* xr = dx + (long)idx[k] * d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < d4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c);
                v += dv[j] * a;
                *reinterpret_cast<f32x4*>(xr + 4 * c) = v;
            }
        }
    }
}

// partial[b][n] = sum over the 64 rows of block b of da[r] T[r][n]   (then launch_colsum_add: fixed order)
constexpr int SEGQ_ROWS = 64;
__global__ __launch_bounds__(256) void seg_dq_kernel(long n_rows, int q, const float* da, const float* T, float* partial) {
    __shared__ float sda[SEGQ_ROWS];
    const long r0 = (long)blockIdx.x * SEGQ_ROWS;
    const int nr = (int)(r0 + SEGQ_ROWS < n_rows ? SEGQ_ROWS : n_rows - r0);
    if (threadIdx.x < SEGQ_ROWS) sda[threadIdx.x] = threadIdx.x < nr ? da[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    for (int n = threadIdx.x; n < q; n += 256) {
        const float* t = T + r0 * q + n;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = 0;
        for (; r + 4 <= nr; r += 4) {
            a0 += sda[r] * t[(long)r * q];
            a1 += sda[r + 1] * t[(long)(r + 1) * q];
            a2 += sda[r + 2] * t[(long)(r + 2) * q];
            a3 += sda[r + 3] * t[(long)(r + 3) * q];
        }
        for (; r < nr; ++r) a0 += sda[r] * t[(long)r * q];
        partial[(long)blockIdx.x * q + n] = (a0 + a1) + (a2 + a3);
    }
}

// ---- index lists from padded neighbour lists: lists [n_seg, K] int64, an entry outside [0, n_rows) = no neighbour (e.g. -1)
//      -> seg_ptr [n_seg + 1], idx (a list's live entries in their order).  Three launches: count -> one-workgroup scan -> emit.
__global__ __launch_bounds__(256) void csr_count_kernel(long n_seg, int K, const int64_t* lists, long n_rows, int* seg_ptr) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s == 0) seg_ptr[0] = 0;
    if (s >= n_seg) return;
    int c = 0;
    for (int k = 0; k < K; ++k) { const int64_t v = lists[s * K + k]; c += (v >= 0 && v < n_rows) ? 1 : 0; }
    seg_ptr[s + 1] = c;
}

// in-place inclusive scan of p[1 .. n]: thread t owns the contiguous chunk [t * chunk, (t + 1) * chunk)
__global__ __launch_bounds__(1024) void csr_scan_kernel(long n, int* p) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const long chunk = (n + 1023) / 1024;
    const long i0 = 1 + t * chunk, i1 = (i0 + chunk < n + 1) ? i0 + chunk : n + 1;
    int sum = 0;
    for (long i = i0; i < i1; ++i) sum += p[i];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                     // Hillis-Steele over the 1024 chunk sums
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;                                 // exclusive prefix of this chunk
    for (long i = i0; i < i1; ++i) { run += p[i]; p[i] = run; }
}

__global__ __launch_bounds__(256) void csr_emit_kernel(long n_seg, int K, const int64_t* lists, long n_rows, const int* seg_ptr, int* idx) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_seg) return;
    int o = seg_ptr[s];
    for (int k = 0; k < K; ++k) { const int64_t v = lists[s * K + k]; if (v >= 0 && v < n_rows) idx[o++] = (int)v; }
}

struct SegWs { size_t wplanes, da, dal, wadd_t, tn_partial, dq_partial, seg_of, key, val, key2, val2, tptr, spart, sort_tmp, sort_tmp_bytes, total; long cap; };
static int key_bits(long n_rows) { int b = 1; while ((1L << b) <= n_rows) ++b; return b; }
static size_t up256(size_t x) { return (x + 255) / 256 * 256; }
static SegWs seg_layout(const nrms_segpool_desc* d) {
    SegWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up256(off + bytes); return o; };
    const size_t pl = d->precision == NRMS_PRECISION_FP32 ? 0 : std::max(gemm_nt_bf16_wplane_bytes(d->q, d->d), gemm_nt_bf16_wplane_bytes(d->d, d->q));
    w.wplanes = take(pl);
    w.da = take((size_t)d->n_rows * 4);
    w.dal = take((size_t)d->nnz * 4);
    w.wadd_t = take((size_t)d->q * d->d * 4);
    w.tn_partial = take(gemm_tn_workspace_floats((int)d->n_rows, d->q, d->d, nullptr) * 4);
    w.dq_partial = take((size_t)cdiv(d->n_rows > 0 ? d->n_rows : 1, SEGQ_ROWS) * d->q * 4);
    w.seg_of = w.key = w.val = w.key2 = w.val2 = w.tptr = w.spart = w.sort_tmp = w.sort_tmp_bytes = 0;
    w.cap = 0;
    if ((d->flags & NRMS_SEGPOOL_ROWS_UNIQUE) == 0 && d->nnz > 0) {
        w.cap = (d->nnz + SEG_SPAN - 1) / SEG_SPAN * SEG_SPAN;          // sorted arrays in whole spans (the padding sorts to the end)
        const size_t e = (size_t)w.cap * 4;
        w.seg_of = take(e); w.key = take(e); w.val = take(e); w.key2 = take(e); w.val2 = take(e);
        w.tptr = take(((size_t)d->n_rows + 1) * 4);
        w.spart = take((size_t)(w.cap / SEG_SPAN) * 2 * d->d * 4);
        size_t tb = 0;
        (void)rocprim::radix_sort_pairs(nullptr, tb, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr, (size_t)w.cap, 0,
                                        key_bits(d->n_rows), (hipStream_t)0);
        w.sort_tmp_bytes = tb;
        w.sort_tmp = take(tb);
    }
    w.total = off;
    return w;
}

static int seg_validate(const nrms_segpool_desc* d, const char* who) {
    NRMS_REQUIRE(d != nullptr, "%s: null desc", who);
    NRMS_REQUIRE(d->n_rows >= 0 && d->n_seg >= 0 && d->nnz >= 0 && d->n_rows < (1L << 31) / 1024 * 512 && d->nnz < (1L << 31) && d->n_seg < (1L << 31),
                 "%s: n_rows=%ld n_seg=%ld nnz=%ld", who, (long)d->n_rows, (long)d->n_seg, (long)d->nnz);
    NRMS_REQUIRE(d->d > 0 && (d->d & 3) == 0 && d->d <= 1024, "%s: d=%d must be a positive multiple of 4, <= 1024", who, d->d);
    NRMS_REQUIRE(d->q > 0 && (d->q & 3) == 0 && d->q <= 512, "%s: q=%d must be a positive multiple of 4, <= 512", who, d->q);
    NRMS_REQUIRE((long)d->n_rows * std::max(d->d, d->q) < (1L << 31), "%s: n_rows * max(d, q) overflows int32", who);
    NRMS_REQUIRE(d->precision == NRMS_PRECISION_FP32 || d->precision == NRMS_PRECISION_BF16X3 || d->precision == NRMS_PRECISION_BF16,
                 "%s: precision %d (fp32, bf16x3 or bf16)", who, d->precision);
    NRMS_REQUIRE((d->flags & ~NRMS_SEGPOOL_ROWS_UNIQUE) == 0, "%s: unknown flags 0x%x", who, d->flags);
    return NRMS_OK;
}

}  // namespace nrms

using namespace nrms;

extern "C" size_t nrms_segment_pool_workspace_bytes(const nrms_segpool_desc* desc) {
    if (seg_validate(desc, "segment_pool_workspace_bytes")) return 0;
    return seg_layout(desc).total;
}

extern "C" int nrms_segment_pool_fwd(const nrms_segpool_desc* desc, const float* x, const float* w_add, const float* b_add,
                                     const float* q_vec, const int32_t* seg_ptr, const int32_t* idx, float* t, float* logit,
                                     float* alpha, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = seg_validate(desc, "segment_pool_fwd");
    if (rc) return rc;
    NRMS_REQUIRE(w_add && b_add && q_vec && seg_ptr, "segment_pool_fwd: null argument");
    NRMS_REQUIRE(desc->n_rows == 0 || (x && t && logit), "segment_pool_fwd: x, t, logit are required");
    NRMS_REQUIRE(desc->nnz == 0 || (idx && alpha), "segment_pool_fwd: idx and alpha are required");
    NRMS_REQUIRE(desc->n_seg == 0 || out, "segment_pool_fwd: null out");
    const SegWs L = seg_layout(desc);
    if (workspace_bytes < L.total || (L.total && workspace == nullptr)) { set_error("segment_pool_fwd: workspace %zu < required %zu bytes", workspace_bytes, L.total); return NRMS_EWORKSPACE; }
    NRMS_REQUIRE(((uintptr_t)workspace & 255) == 0, "segment_pool_fwd: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int d = desc->d, q = desc->q;
    if (desc->n_rows > 0) {
        NTArgs g{};
        g.M = (int)desc->n_rows; g.N = q; g.K = d; g.rows_per_tile = NT_BM;
        g.A = x; g.lda = d; g.W = w_add; g.bias = b_add; g.C = t; g.ldc = q;
        if (desc->precision == NRMS_PRECISION_FP32) rc = launch_gemm_nt(A_PLAIN, E_STORE, g, s, "segpool_proj_fwd");
        else rc = launch_gemm_nt_bf16(A_PLAIN, E_STORE, desc->precision == NRMS_PRECISION_BF16X3 ? 3 : 1, g, (char*)workspace + L.wplanes, s, "segpool_proj_fwd");
        if (rc) return rc;
        TimingScope ts("segpool_logit", s);
        hipLaunchKernelGGL(seg_logit_kernel, dim3(cdiv(desc->n_rows, SEG_WPB)), dim3(64 * SEG_WPB), 0, s, (long)desc->n_rows, q, t, q_vec, logit);
        rc = check_launch("segpool_logit");
        if (rc) return rc;
    }
    if (desc->n_seg == 0) return NRMS_OK;
    TimingScope ts("segpool_fwd", s);
    hipLaunchKernelGGL(seg_pool_fwd_kernel, dim3(cdiv(desc->n_seg, SEG_WPB)), dim3(64 * SEG_WPB), 0, s, (long)desc->n_seg, d, x, seg_ptr, idx,
                       logit, alpha, out);
    return check_launch("segpool_fwd");
}

extern "C" int nrms_segment_pool_bwd(const nrms_segpool_desc* desc, const float* x, const float* w_add, const float* q_vec,
                                     const int32_t* seg_ptr, const int32_t* idx, const float* t, const float* alpha,
                                     const float* dout, float* dx, float* dw_add, float* db_add, float* dq_vec, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    int rc = seg_validate(desc, "segment_pool_bwd");
    if (rc) return rc;
    NRMS_REQUIRE(w_add && q_vec && seg_ptr && dw_add && db_add && dq_vec, "segment_pool_bwd: null argument");
    NRMS_REQUIRE(desc->n_rows == 0 || (x && t && dx), "segment_pool_bwd: x, t, dx are required");
    NRMS_REQUIRE(desc->nnz == 0 || (idx && alpha), "segment_pool_bwd: idx and alpha are required");
    NRMS_REQUIRE(desc->n_seg == 0 || dout, "segment_pool_bwd: null dout");
    const SegWs L = seg_layout(desc);
    if (workspace_bytes < L.total || (L.total && workspace == nullptr)) { set_error("segment_pool_bwd: workspace %zu < required %zu bytes", workspace_bytes, L.total); return NRMS_EWORKSPACE; }
    NRMS_REQUIRE(((uintptr_t)workspace & 255) == 0, "segment_pool_bwd: workspace must be 256-byte aligned");
    if (desc->n_rows == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int d = desc->d, q = desc->q;
    const bool unique = (desc->flags & NRMS_SEGPOOL_ROWS_UNIQUE) != 0;
    const bool by_row = !unique && desc->nnz > 0 && desc->n_seg > 0;
    char* base = (char*)workspace;
    int* seg_of = (int*)(base + L.seg_of); int* tval = (int*)(base + L.val2); int* tptr = (int*)(base + L.tptr);
    float* da = (float*)(base + L.da);
    float* dal = (float*)(base + L.dal);
    float* wadd_t = (float*)(base + L.wadd_t);
    // rows that are nobody's member get d(logit) = 0
    if (hipMemsetAsync(da, 0, (size_t)desc->n_rows * 4, s) != hipSuccess) { set_error("segment_pool_bwd: hipMemsetAsync failed"); return NRMS_ELAUNCH; }
    const dim3 sgrid(cdiv(desc->n_seg > 0 ? desc->n_seg : 1, SEG_WPB)), sblk(64 * SEG_WPB);
    if (desc->n_seg > 0) {
        TimingScope ts("segpool_da", s);
        if (unique) hipLaunchKernelGGL(seg_da_kernel<true>, sgrid, sblk, 0, s, (long)desc->n_seg, d, x, seg_ptr, idx, alpha, dout, dal, da, (int*)nullptr);
        else hipLaunchKernelGGL(seg_da_kernel<false>, sgrid, sblk, 0, s, (long)desc->n_seg, d, x, seg_ptr, idx, alpha, dout, dal, da, seg_of);
        rc = check_launch("segpool_da");
        if (rc) return rc;
    }
    if (by_row) {
        // the entries of every row, in ascending list position: stable sort of (row, entry) pairs, then the rows' bounds
        TimingScope ts("segpool_rowlists", s);
        const long cap = L.cap;
        int* key = (int*)(base + L.key); int* val = (int*)(base + L.val); int* key2 = (int*)(base + L.key2);
        hipLaunchKernelGGL(seg_tkey_kernel, dim3(cdiv(cap, 256)), dim3(256), 0, s, cap, idx, seg_ptr + desc->n_seg, (int)desc->n_rows, key, val);
        size_t tb = L.sort_tmp_bytes;
        if (rocprim::radix_sort_pairs(base + L.sort_tmp, tb, (const int*)key, key2, (const int*)val, tval, (size_t)cap, 0, key_bits(desc->n_rows), s) != hipSuccess) {
            set_error("segment_pool_bwd: radix_sort_pairs failed");
            return NRMS_ELAUNCH;
        }
        hipLaunchKernelGGL(seg_tptr_kernel, dim3(cdiv(desc->n_rows + 1, 256)), dim3(256), 0, s, cap, (int)desc->n_rows, (const int*)key2, tptr);
        hipLaunchKernelGGL(seg_row_da_kernel, dim3(cdiv(desc->n_rows, SEG_WPB)), dim3(64 * SEG_WPB), 0, s, (long)desc->n_rows, (const int*)tptr, (const int*)tval,
                           (const float*)dal, da);
        rc = check_launch("segpool_rowlists");
        if (rc) return rc;
    }
    // d(q_vec) += sum_r da_r T_r
    {
        float* part = (float*)(base + L.dq_partial);
        const int nb = cdiv(desc->n_rows, SEGQ_ROWS);
        {
            TimingScope ts("segpool_dq", s);
            hipLaunchKernelGGL(seg_dq_kernel, dim3(nb), dim3(256), 0, s, (long)desc->n_rows, q, da, t, part);
        }
        rc = check_launch("segpool_dq");
        if (rc) return rc;
        rc = launch_colsum_add(part, nb, q, dq_vec, s);
        if (rc) return rc;
    }
    // d(W_add), d(b_add) += dZ^T [x | 1],  dZ = da q_vec (1 - T^2)
    {
        TNArgs tn{};
        tn.M = (int)desc->n_rows; tn.N = q; tn.K = d; tn.amode = A_DZ;
        tn.ds = da; tn.qv = q_vec; tn.T = t; tn.B = x; tn.ldb = d;
        tn.dW = dw_add; tn.dbias = db_add; tn.partial = (float*)(base + L.tn_partial);
        rc = desc->precision == NRMS_PRECISION_FP32 ? launch_gemm_tn(tn, s, "segpool_dwadd")
                                                    : launch_gemm_tn_bf16(desc->precision == NRMS_PRECISION_BF16X3 ? 3 : 1, tn, s, "segpool_dwadd");
        if (rc) return rc;
    }
    // dx = dZ W_add (overwrites), then + alpha_k dout_s through the index lists
    rc = launch_transpose(w_add, wadd_t, q, d, s);
    if (rc) return rc;
    {
        NTArgs g{};
        g.M = (int)desc->n_rows; g.N = d; g.K = q; g.rows_per_tile = NT_BM;
        g.ds = da; g.qv = q_vec; g.T = t; g.W = wadd_t; g.C = dx; g.ldc = d;
        if (desc->precision == NRMS_PRECISION_FP32) rc = launch_gemm_nt(A_DZ, E_STORE, g, s, "segpool_dx");
        else rc = launch_gemm_nt_bf16(A_DZ, E_STORE, desc->precision == NRMS_PRECISION_BF16X3 ? 3 : 1, g, base + L.wplanes, s, "segpool_dx");
        if (rc) return rc;
    }
    if (desc->n_seg > 0) {
        TimingScope ts("segpool_scatter", s);
        if (unique) hipLaunchKernelGGL(seg_scatter_kernel, sgrid, sblk, 0, s, (long)desc->n_seg, d, seg_ptr, idx, alpha, dout, dx);
        else if (by_row) {
            const long n_spans = L.cap / SEG_SPAN;
            float* spart = (float*)(base + L.spart);
            hipLaunchKernelGGL(seg_span_gather_kernel, dim3(cdiv(n_spans, SEG_WPB)), dim3(64 * SEG_WPB), 0, s, n_spans, (int)desc->n_rows, d,
                               (const int*)(base + L.key2), (const int*)tval, (const int*)tptr, (const int*)seg_of, alpha, dout, spart, dx);
            hipLaunchKernelGGL(seg_row_combine_kernel, dim3(cdiv(desc->n_rows, SEG_WPB)), dim3(64 * SEG_WPB), 0, s, (long)desc->n_rows, d, (const int*)tptr,
                               (const float*)spart, dx);
        }
        rc = check_launch("segpool_scatter");
    }
    return rc;
}

extern "C" int nrms_csr_from_padded(int64_t n_seg, int32_t K, const int64_t* lists, int64_t n_rows, int32_t* seg_ptr, int32_t* idx,
                                    void* stream) {
    NRMS_REQUIRE(n_seg >= 0 && K > 0 && n_rows >= 0 && n_rows < (1L << 31) && n_seg * (long)K < (1L << 31),
                 "csr_from_padded: n_seg=%ld K=%d n_rows=%ld", (long)n_seg, K, (long)n_rows);
    NRMS_REQUIRE(seg_ptr != nullptr, "csr_from_padded: null seg_ptr");
    NRMS_REQUIRE(n_seg == 0 || (lists && idx), "csr_from_padded: null argument");
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("csr_build", s);
    hipLaunchKernelGGL(csr_count_kernel, dim3(cdiv(n_seg > 0 ? n_seg : 1, 256)), dim3(256), 0, s, (long)n_seg, K, lists, (long)n_rows, seg_ptr);
    if (n_seg > 0) {
        hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, s, (long)n_seg, seg_ptr);
        hipLaunchKernelGGL(csr_emit_kernel, dim3(cdiv(n_seg, 256)), dim3(256), 0, s, (long)n_seg, K, lists, (long)n_rows, seg_ptr, idx);
    }
    return check_launch("csr_from_padded");
}
