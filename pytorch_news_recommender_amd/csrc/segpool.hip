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
//   backward: d(alpha_k) = <dout_s, x_k>, d(logit) by the softmax rule    (seg_da: plain stores when the segments partition the
//             rows, float atomics otherwise), d(q_vec) = sum_r d(logit_r) T_r (fixed-order partial sums),
//             d(W_add), d(b_add) = dZ^T [x | 1] and dx = dZ W_add with dZ = d(logit) q_vec (1 - T^2)   (the TN / NT GEMMs' dZ loaders),
//             dx[idx_k] += alpha_k dout_s                                  (seg_scatter)
#include <algorithm>

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

// d(logit) of every member: da_k = alpha_k (<dout_s, x_k> - sum_j alpha_j <dout_s, x_j>), into da_row[idx_k]
// (UNIQUE: a row has one membership at most -> plain store; otherwise atomic add into the zeroed buffer).  dal [nnz]: scratch.
template <bool UNIQUE>
__global__ __launch_bounds__(64 * SEG_WPB) void seg_da_kernel(long n_seg, int d, const float* x, const int* seg_ptr, const int* idx,
                                                              const float* alpha, const float* dout, float* dal, float* da_row) {
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
        else atomicAdd(da_row + idx[k], da);
    }
}

// dx[idx_k][:] += alpha_k dout_s[:]
template <bool UNIQUE>
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
                if (UNIQUE) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c);
                    v += dv[j] * a;
                    *reinterpret_cast<f32x4*>(xr + 4 * c) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(xr + 4 * c + e, a * dv[j][e]);
                }
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

struct SegWs { size_t wplanes, da, dal, wadd_t, tn_partial, dq_partial, total; };
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
    char* base = (char*)workspace;
    float* da = (float*)(base + L.da);
    float* dal = (float*)(base + L.dal);
    float* wadd_t = (float*)(base + L.wadd_t);
    // rows that are nobody's member get d(logit) = 0
    if (hipMemsetAsync(da, 0, (size_t)desc->n_rows * 4, s) != hipSuccess) { set_error("segment_pool_bwd: hipMemsetAsync failed"); return NRMS_ELAUNCH; }
    const dim3 sgrid(cdiv(desc->n_seg > 0 ? desc->n_seg : 1, SEG_WPB)), sblk(64 * SEG_WPB);
    if (desc->n_seg > 0) {
        TimingScope ts("segpool_da", s);
        if (unique) hipLaunchKernelGGL(seg_da_kernel<true>, sgrid, sblk, 0, s, (long)desc->n_seg, d, x, seg_ptr, idx, alpha, dout, dal, da);
        else hipLaunchKernelGGL(seg_da_kernel<false>, sgrid, sblk, 0, s, (long)desc->n_seg, d, x, seg_ptr, idx, alpha, dout, dal, da);
        rc = check_launch("segpool_da");
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
        if (unique) hipLaunchKernelGGL(seg_scatter_kernel<true>, sgrid, sblk, 0, s, (long)desc->n_seg, d, seg_ptr, idx, alpha, dout, dx);
        else hipLaunchKernelGGL(seg_scatter_kernel<false>, sgrid, sblk, 0, s, (long)desc->n_seg, d, seg_ptr, idx, alpha, dout, dx);
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
