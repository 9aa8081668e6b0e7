// Shape-general fp32 kernels for the variant encoders (SURVEY f-3: nrms_naml, /root/reference/MIND_2020/model/nrms_naml.py):
//   * scaled-dot-product attention for any d_k <= 128 with dropout on the attention PROBABILITIES
//     (Attention.forward, nrms_naml.py:20-41) -- the MFMA attention kernels (attention.hip) cover even d_k <= 64 and
//     drop the context instead, as nrms_v0 does;
//   * the row part of the additive attention for q_dim > 256 or d_model > 512 (AdditiveAttention, nrms_naml.py:77-100;
//     the user encoder of nrms_naml is 800 wide with a 400-wide query, config.py:68,72);
//   * LayerNorm over the history vectors (nrms_naml.py:207,238);
//   * the news feature row [title | abstract | category | sub-category] with its dropout (nrms_naml.py:170-175) and
//     the category-table gradients.
// These run on the VALU: together they are a few percent of a step's flops (the projections stay on the MFMA GEMMs),
// and they have to take any width the config names rather than a tile shape.
#include "gemm.h"

namespace nrms {

static __device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =======================================================================================
// Attention: one workgroup (4 waves) per (sequence, head).  Q, K, V (and dO) rows and the S x S matrices live in LDS;
// every product is a small matrix multiply done in 4 x 4 register blocks per thread (16 FMAs per 8 LDS reads), the
// softmax rows are dealt to the waves.
// =======================================================================================
struct WideAttn {
    int n_seq, S, d, h, dk, dkp, sp;      // dkp / sp: LDS row pitches (floats, odd) of the [S][dk] operands / the [S][S] matrices
    float scale;
    const float* qkv;       // [M, 3d] head-major columns [head][Q | K | V][dk] (HeadPerm, common.h)
    float* ctx;             // fwd out [M, d]
    const float* dctx;      // bwd in  [M, d]
    float* dqkv;            // bwd out [M, 3d], head-major
    const uint8_t* mask;    // optional [n_seq, S]: pair (i, j) is masked unless mask_i and mask_j
    Dropout pdrop;          // site 2, element ((seq * h + head) * S + i) * S + j
};

// C(m, n) = sum_k A[m * am + k * ak] * B[n * bn + k * bk] for m < M, n < N, handed to `emit(m, n, value)`.
// A thread owns rows m0..m0+3 and the columns {n0 + v * NB}: consecutive lanes walk consecutive n (odd pitches and
// unit strides are both conflict-free), lanes of one row block share their A addresses (LDS broadcast).
template <class Emit>
static __device__ __forceinline__ void block_mm(const float* A, int am, int ak, const float* B, int bn, int bk, int M, int N,
                                                int K, Emit emit) {
    const int MB = (M + 3) >> 2, NB = (N + 3) >> 2;
    for (int blk = threadIdx.x; blk < MB * NB; blk += blockDim.x) {
        const int m0 = (blk / NB) * 4, n0 = blk % NB;
        const float* ap[4];
        const float* bp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ap[u] = A + min(m0 + u, M - 1) * am;
            bp[u] = B + min(n0 + u * NB, N - 1) * bn;
        }
        float acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = 0.f;
#pragma unroll 2
        for (int k = 0; k < K; ++k) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = ap[u][k * ak]; bv[u] = bp[u][k * bk]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += av[u] * bv[v];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (m0 + u < M && n0 + v * NB < N) emit(m0 + u, n0 + v * NB, acc[u][v]);
    }
}

template <bool BWD>
__global__ __launch_bounds__(256) void attn_wide_kernel(WideAttn a) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S, dk = a.dk, dkp = a.dkp, sp = a.sp;
    float* Q = wsm;
    float* K = Q + S * dkp;
    float* V = K + S * dkp;
    float* Pd = V + S * dkp;               // scores -> dropped probabilities (fwd); dP -> dS (bwd)
    float* P = Pd + S * sp;                // bwd only: probabilities
    float* dO = P + S * sp;                // bwd only
    const long units = (long)a.n_seq * a.h;
    for (long unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const long seq = unit / a.h;
        const int head = (int)(unit - seq * a.h);
        const long row0 = seq * S;
        for (int idx = threadIdx.x; idx < S * 3 * dk; idx += blockDim.x) {
            const int i = idx / (3 * dk), c = idx - i * 3 * dk;
            const int which = c / dk, j = c - which * dk;
            Q[which * S * dkp + i * dkp + j] = a.qkv[(row0 + i) * 3 * a.d + head * 3 * dk + c];
        }
        if (BWD)
            for (int idx = threadIdx.x; idx < S * dk; idx += blockDim.x) {
                const int i = idx / dk, j = idx - i * dk;
                dO[i * dkp + j] = a.dctx[(row0 + i) * a.d + head * dk + j];
            }
        __syncthreads();
        // scores[i][j] = Q_i . K_j
        float* Sc = BWD ? P : Pd;
        block_mm(Q, dkp, 1, K, dkp, 1, S, S, dk, [&](int i, int j, float v) { Sc[i * sp + j] = v; });
        __syncthreads();
        // softmax over the keys of each query row (one wave per row, lane = key), then the probability dropout
        const bool jok = lane < S;
        const bool mj = a.mask == nullptr || (jok && a.mask[row0 + lane] != 0);
        for (int i = wave; i < S; i += 4) {
            float s = jok ? Sc[i * sp + lane] * a.scale : -3.0e38f;
            if (jok && a.mask != nullptr && !(mj && a.mask[row0 + i] != 0)) s = -1e9f;
            const float mx = wave_max(s);
            const float e = jok ? expf(s - mx) : 0.f;
            const float p = e / wave_sum(e);
            float keep = 1.f;
            if (a.pdrop.thresh != 0u)
                keep = dropout_scale1(a.pdrop.seed, 2u, (uint64_t)((unit * S + i) * S + lane), a.pdrop.thresh, a.pdrop.inv_keep);
            if (jok) {
                Pd[i * sp + lane] = p * keep;
                if (BWD) P[i * sp + lane] = p;
            }
        }
        __syncthreads();
        if (!BWD) {
            // ctx[i][c] = sum_j Pd[i][j] V[j][c]
            float* out = a.ctx + row0 * a.d + head * dk;
            block_mm(Pd, sp, 1, V, 1, dkp, S, dk, S, [&](int i, int c, float v) { out[(long)i * a.d + c] = v; });
        } else {
            float* out = a.dqkv + row0 * 3 * a.d + head * 3 * dk;
            // dV[j][c] = sum_i Pd[i][j] dO[i][c]
            block_mm(Pd, 1, sp, dO, 1, dkp, S, dk, S, [&](int j, int c, float v) { out[(long)j * 3 * a.d + 2 * dk + c] = v; });
            __syncthreads();
            // dPd[i][j] = dO_i . V_j  (overwrites Pd)
            block_mm(dO, dkp, 1, V, dkp, 1, S, S, dk, [&](int i, int j, float v) { Pd[i * sp + j] = v; });
            __syncthreads();
            // dS = P (dP - sum_j dP P) * scale, dP = dPd * keep
            for (int i = wave; i < S; i += 4) {
                float dp = jok ? Pd[i * sp + lane] : 0.f;
                if (a.pdrop.thresh != 0u)
                    dp *= dropout_scale1(a.pdrop.seed, 2u, (uint64_t)((unit * S + i) * S + lane), a.pdrop.thresh, a.pdrop.inv_keep);
                const float p = jok ? P[i * sp + lane] : 0.f;
                const float delta = wave_sum(dp * p);
                float ds = p * (dp - delta) * a.scale;
                if (a.mask != nullptr && !(mj && a.mask[row0 + i] != 0)) ds = 0.f;     // masked_fill passes no gradient
                if (jok) Pd[i * sp + lane] = ds;
            }
            __syncthreads();
            // dQ[i][c] = sum_j dS[i][j] K[j][c];  dK[j][c] = sum_i dS[i][j] Q[i][c]
            block_mm(Pd, sp, 1, K, 1, dkp, S, dk, S, [&](int i, int c, float v) { out[(long)i * 3 * a.d + c] = v; });
            block_mm(Pd, 1, sp, Q, 1, dkp, S, dk, S, [&](int j, int c, float v) { out[(long)j * 3 * a.d + dk + c] = v; });
        }
        __syncthreads();
    }
}

int launch_attention_wide(bool bwd, int n_seq, int S, int d, int h, const float* qkv, float* ctx, const Dropout& pdrop,
                          const float* dctx, float* dqkv, const uint8_t* mask, hipStream_t stream) {
    if (n_seq <= 0) return NRMS_OK;
    WideAttn a{};
    a.n_seq = n_seq; a.S = S; a.d = d; a.h = h; a.dk = d / h;
    if (S < 1 || S > 64 || a.dk < 1 || a.dk > 128) {
        set_error("attention (wide): unsupported S=%d d_k=%d (need 1<=S<=64, d_k<=128)", S, a.dk);
        return NRMS_EINVAL;
    }
    a.dkp = a.dk | 1;                      // odd pitches: a lane per row of one column hits distinct banks
    a.sp = S | 1;
    a.scale = 1.0f / sqrtf((float)a.dk);
    a.qkv = qkv; a.ctx = ctx; a.dctx = dctx; a.dqkv = dqkv; a.mask = mask; a.pdrop = pdrop;
    const size_t lds = (size_t)((bwd ? 4 : 3) * S * a.dkp + (bwd ? 2 : 1) * S * a.sp) * sizeof(float);
    const void* fn = bwd ? (const void*)attn_wide_kernel<true> : (const void*)attn_wide_kernel<false>;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("attention (wide): hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    long blocks = (long)n_seq * h;
    if (blocks > 256 * 64) blocks = 256 * 64;
    TimingScope ts(bwd ? "wide_attn_bwd" : "wide_attn_fwd", stream);
    if (bwd) hipLaunchKernelGGL(attn_wide_kernel<true>, dim3((int)blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(attn_wide_kernel<false>, dim3((int)blocks), dim3(256), lds, stream, a);
    return check_launch("attention_wide");
}

// =======================================================================================
// Additive attention, row part, any width.  One workgroup (4 waves) per sequence.
//   forward : Z [S, q] = ctx Wa^T + ba arrives from the NT GEMM in the T buffer; T = tanh(Z) in place,
//             s_t = T_t . q_vec, w = softmax_S(s) (optional mask -> -1e9), out = sum_t w_t ctx_t
//   backward: dw_t = ctx_t . dout, ds_t = w_t (dw_t - sum w dw), d(q_vec) partial row = sum_t ds_t T_t
// =======================================================================================
__global__ __launch_bounds__(256) void addattn_rows_fwd_wide_kernel(int n_seq, int S, int d, int q, float* T, const float* qv,
                                                                    const float* ctx, const uint8_t* mask, float* wout,
                                                                    float* out) {
    __shared__ float sc[64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (long seq = blockIdx.x; seq < n_seq; seq += gridDim.x) {
        for (int t = wave; t < S; t += 4) {
            float* z = T + (seq * S + t) * q;
            float part = 0.f;
            for (int n = lane; n < q; n += 64) {
                const float tv = tanhf(z[n]);
                z[n] = tv;
                part += tv * qv[n];
            }
            part = wave_sum(part);
            if (lane == 0) sc[t] = (mask != nullptr && mask[seq * S + t] == 0) ? -1e9f : part;
        }
        __syncthreads();
        if (wave == 0) {
            const float v = lane < S ? sc[lane] : -3.0e38f;
            const float mx = wave_max(v);
            const float e = lane < S ? expf(v - mx) : 0.f;
            const float w = e / wave_sum(e);
            if (lane < S) {
                sc[lane] = w;
                if (wout != nullptr) wout[seq * S + lane] = w;
            }
        }
        __syncthreads();
        const float* c0 = ctx + seq * S * d;
        for (int c = threadIdx.x; c < d; c += 256) {
            float o = 0.f;
            for (int t = 0; t < S; ++t) o += sc[t] * c0[(long)t * d + c];
            out[seq * d + c] = o;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void addattn_rows_bwd_wide_kernel(int n_seq, int S, int d, int q, const float* ctx,
                                                                    const float* dout, const float* w, const float* T,
                                                                    float* ds, float* dq_partial, const uint8_t* mask) {
    __shared__ float dwl[64], dsl[64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (long seq = blockIdx.x; seq < n_seq; seq += gridDim.x) {
        const float* g = dout + seq * d;
        for (int t = wave; t < S; t += 4) {
            const float* c = ctx + (seq * S + t) * d;
            float p = 0.f;
            for (int k = lane; k < d; k += 64) p += c[k] * g[k];
            p = wave_sum(p);
            if (lane == 0) dwl[t] = p;
        }
        __syncthreads();
        if (wave == 0) {
            const float my_dw = lane < S ? dwl[lane] : 0.f;
            const float my_w = lane < S ? w[seq * S + lane] : 0.f;
            const float dot = wave_sum(my_w * my_dw);
            float my_ds = my_w * (my_dw - dot);
            if (mask != nullptr && lane < S && mask[seq * S + lane] == 0) my_ds = 0.f;
            if (lane < S) { ds[seq * S + lane] = my_ds; dsl[lane] = my_ds; }
        }
        __syncthreads();
        const float* t0 = T + seq * S * q;
        for (int n = threadIdx.x; n < q; n += 256) {
            float acc = 0.f;
            for (int t = 0; t < S; ++t) acc += dsl[t] * t0[(long)t * q + n];
            dq_partial[seq * q + n] = acc;                              // one partial row per sequence
        }
        __syncthreads();
    }
}

int launch_addattn_rows_fwd_wide(int n_seq, int S, int d, int q, float* T, const float* q_vec, const float* ctx,
                                 const uint8_t* mask, float* wout, float* out, hipStream_t stream) {
    if (n_seq <= 0) return NRMS_OK;
    if (S > 64) { set_error("addattn (wide): S=%d > 64", S); return NRMS_EINVAL; }
    TimingScope ts("addattn_fwd", stream);
    hipLaunchKernelGGL(addattn_rows_fwd_wide_kernel, dim3(n_seq > 8192 ? 8192 : n_seq), dim3(256), 0, stream, n_seq, S, d, q, T,
                       q_vec, ctx, mask, wout, out);
    return check_launch("addattn_rows_fwd_wide");
}

// dq_partial: n_seq rows of q floats (addattn_bwd_rows_partial_rows)
int launch_addattn_rows_bwd_wide(int n_seq, int S, int d, int q, const float* ctx, const float* dout, const float* w,
                                 const float* T, float* ds, float* dq_partial, float* dq, const uint8_t* mask,
                                 hipStream_t stream) {
    if (n_seq <= 0) return NRMS_OK;
    if (S > 64) { set_error("addattn (wide): S=%d > 64", S); return NRMS_EINVAL; }
    {
        TimingScope ts("addattn_bwd_rows", stream);
        hipLaunchKernelGGL(addattn_rows_bwd_wide_kernel, dim3(n_seq), dim3(256), 0, stream, n_seq, S, d, q, ctx, dout, w, T, ds,
                           dq_partial, mask);
        int rc = check_launch("addattn_rows_bwd_wide");
        if (rc) return rc;
    }
    return launch_colsum_add(dq_partial, n_seq, q, dq, stream);
}

// =======================================================================================
// LayerNorm over the last dimension (torch.nn.LayerNorm, biased variance, nrms_naml.py:207,238).  One wave per row.
// =======================================================================================
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(long n_rows, int d, const float* x, const float* gamma,
                                                            const float* beta, float eps, float* y, float* stats) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* xr = x + row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c];
    const float mean = wave_sum(s) / d;
    float v = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = xr[c] - mean; v += t * t; }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / d + eps);
    for (int c = lane; c < d; c += 64) y[row * d + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
    if (lane == 0 && stats != nullptr) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(long n_rows, int d, const float* x, const float* gamma,
                                                               const float* stats, const float* dy, float* dx) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float* xr = x + row * d;
    const float* gr = dy + row * d;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float g = gr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
        s1 += g;
        s2 += g * xh;
    }
    s1 = wave_sum(s1) / d;
    s2 = wave_sum(s2) / d;
    for (int c = lane; c < d; c += 64) {
        const float g = gr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
        dx[row * d + c] = rstd * (g - s1 - xh * s2);
    }
}

// partial[chunk][0][c] = sum over the chunk's rows of dy xhat, partial[chunk][1][c] = sum of dy  (fixed order)
constexpr int LN_CHUNKS = 128;
__global__ __launch_bounds__(256) void layernorm_bwd_affine_kernel(long n_rows, int d, const float* x, const float* stats,
                                                                   const float* dy, float* partial) {
    const int chunk = blockIdx.x;
    const long per = (n_rows + LN_CHUNKS - 1) / LN_CHUNKS;
    const long r0 = chunk * per, r1 = min(n_rows, r0 + per);
    for (int c = threadIdx.x; c < d; c += 256) {
        float sg = 0.f, sb = 0.f;
        for (long r = r0; r < r1; ++r) {
            const float g = dy[r * d + c];
            sg += g * (x[r * d + c] - stats[2 * r]) * stats[2 * r + 1];
            sb += g;
        }
        partial[(long)chunk * 2 * d + c] = sg;
        partial[(long)chunk * 2 * d + d + c] = sb;
    }
}

int launch_layernorm_fwd(long n_rows, int d, const float* x, const float* gamma, const float* beta, float eps, float* y,
                         float* stats, hipStream_t stream) {
    if (n_rows <= 0) return NRMS_OK;
    TimingScope ts("layernorm_fwd", stream);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((int)cdiv(n_rows, 4)), dim3(256), 0, stream, n_rows, d, x, gamma, beta, eps, y,
                       stats);
    return check_launch("layernorm_fwd");
}

size_t layernorm_bwd_workspace_floats(int d) { return (size_t)LN_CHUNKS * 2 * d; }

// dgamma / dbeta are ACCUMULATED into (as every parameter gradient of this library); they must be adjacent:
// dgamma [d] immediately followed by dbeta [d]
int launch_layernorm_bwd(long n_rows, int d, const float* x, const float* gamma, const float* stats, const float* dy,
                         float* dx, float* dgamma_dbeta, float* workspace, hipStream_t stream) {
    if (n_rows <= 0) return NRMS_OK;
    TimingScope ts("layernorm_bwd", stream);
    hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3((int)cdiv(n_rows, 4)), dim3(256), 0, stream, n_rows, d, x, gamma, stats, dy, dx);
    hipLaunchKernelGGL(layernorm_bwd_affine_kernel, dim3(LN_CHUNKS), dim3(256), 0, stream, n_rows, d, x, stats, dy, workspace);
    int rc = check_launch("layernorm_bwd");
    if (rc) return rc;
    return launch_colsum_add(workspace, LN_CHUNKS, 2 * d, dgamma_dbeta, stream);
}

// =======================================================================================
// News feature rows (NewsEncoder.forward, nrms_naml.py:168-175):
//   out[n] = dropout([title_vec[n] | abst_vec[n] | cat_table[categ[n]] | sub_table[subcateg[n]]])   (site 3)
// Backward: the text-vector gradients are slices of dout through the same mask; a category-table row is summed over the
// slots that name it by one workgroup in ascending slot order (reproducible, no atomics; row 0 = padding_idx gets none).
// =======================================================================================
struct FeatArgs {
    long n;
    int dt, dc, n_cat, n_sub;
    const float *title, *abst, *cat_table, *sub_table;
    const int64_t *categ, *subcateg;
    Dropout drop;
    float* out;              // fwd
    const float* dout;       // bwd
    float *d_title, *d_abst, *d_cat_table, *d_sub_table;
};

__global__ __launch_bounds__(256) void features_fwd_kernel(FeatArgs a) {
    const int F = 2 * a.dt + 2 * a.dc;
    for (long n = blockIdx.x; n < a.n; n += gridDim.x) {
        long ci = a.categ[n], si = a.subcateg[n];
        if (ci < 0 || ci >= a.n_cat) ci = 0;                            // ids are validated by the caller; never index outside
        if (si < 0 || si >= a.n_sub) si = 0;
        for (int c = threadIdx.x; c < F; c += blockDim.x) {
            float v;
            if (c < a.dt) v = a.title[n * a.dt + c];
            else if (c < 2 * a.dt) v = a.abst[n * a.dt + c - a.dt];
            else if (c < 2 * a.dt + a.dc) v = a.cat_table[ci * a.dc + c - 2 * a.dt];
            else v = a.sub_table[si * a.dc + c - 2 * a.dt - a.dc];
            if (a.drop.thresh != 0u) v *= dropout_scale1(a.drop.seed, 3u, (uint64_t)(n * F + c), a.drop.thresh, a.drop.inv_keep);
            a.out[n * F + c] = v;
        }
    }
}

__global__ __launch_bounds__(256) void features_bwd_text_kernel(FeatArgs a) {
    const int F = 2 * a.dt + 2 * a.dc;
    for (long n = blockIdx.x; n < a.n; n += gridDim.x)
        for (int c = threadIdx.x; c < 2 * a.dt; c += blockDim.x) {
            float g = a.dout[n * F + c];
            if (a.drop.thresh != 0u) g *= dropout_scale1(a.drop.seed, 3u, (uint64_t)(n * F + c), a.drop.thresh, a.drop.inv_keep);
            if (c < a.dt) a.d_title[n * a.dt + c] = g;
            else a.d_abst[n * a.dt + c - a.dt] = g;
        }
}

// block b < n_cat - 1: category row b + 1; otherwise sub-category row b - (n_cat - 1) + 1.  128 threads: a window of
// 128 slots is tested with one id per thread, the two wave ballots name the hits, and every thread (= one column)
// walks the set bits in ascending slot order.
__global__ __launch_bounds__(128) void features_bwd_tables_kernel(FeatArgs a) {
    const int F = 2 * a.dt + 2 * a.dc;
    const bool is_cat = (int)blockIdx.x < a.n_cat - 1;
    const int row = is_cat ? blockIdx.x + 1 : blockIdx.x - (a.n_cat - 1) + 1;
    const int64_t* ids = is_cat ? a.categ : a.subcateg;
    const int col0 = 2 * a.dt + (is_cat ? 0 : a.dc);
    float* dst = (is_cat ? a.d_cat_table : a.d_sub_table) + (long)row * a.dc;
    __shared__ unsigned long long hit[2];
    const int wave = threadIdx.x >> 6;
    for (int c0 = 0; c0 < a.dc; c0 += 128) {
        const int c = c0 + threadIdx.x;
        float acc = 0.f;
        for (long base = 0; base < a.n; base += 128) {
            const long n_mine = base + threadIdx.x;
            const unsigned long long b = __ballot(n_mine < a.n && ids[n_mine] == row);
            __syncthreads();                                            // the previous window's masks are consumed
            if ((threadIdx.x & 63) == 0) hit[wave] = b;
            __syncthreads();
            if (c < a.dc)
                for (int w = 0; w < 2; ++w)
                    for (unsigned long long m = hit[w]; m != 0; m &= m - 1) {
                        const long n = base + 64 * w + __builtin_ctzll(m);
                        float g = a.dout[n * F + col0 + c];
                        if (a.drop.thresh != 0u)
                            g *= dropout_scale1(a.drop.seed, 3u, (uint64_t)(n * F + col0 + c), a.drop.thresh, a.drop.inv_keep);
                        acc += g;
                    }
        }
        if (c < a.dc) dst[c] += acc;
    }
}

int launch_features_fwd(const FeatArgs& a, hipStream_t stream) {
    if (a.n <= 0) return NRMS_OK;
    TimingScope ts("features_fwd", stream);
    hipLaunchKernelGGL(features_fwd_kernel, dim3((int)(a.n > 16384 ? 16384 : a.n)), dim3(256), 0, stream, a);
    return check_launch("features_fwd");
}

// The same sums cut into slot chunks (grid = rows x chunks): partial[chunk][row][:] over the chunk's slots, ascending; the chunks are
// added in ascending order by features_tables_reduce_kernel.  One workgroup per table row walking all 28 160 slots of a 512-user
// batch took 0.45 ms (220 windows of two barriers each).
__global__ __launch_bounds__(128) void features_bwd_tables_chunk_kernel(FeatArgs a, long chunk, float* partial) {
    const int F = 2 * a.dt + 2 * a.dc;
    const bool is_cat = (int)blockIdx.x < a.n_cat - 1;
    const int row = is_cat ? blockIdx.x + 1 : blockIdx.x - (a.n_cat - 1) + 1;
    const int64_t* ids = is_cat ? a.categ : a.subcateg;
    const int col0 = 2 * a.dt + (is_cat ? 0 : a.dc);
    float* dst = partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * a.dc;
    const long s0 = (long)blockIdx.y * chunk, s1 = s0 + chunk < a.n ? s0 + chunk : a.n;
    __shared__ unsigned long long hit[2];
    const int wave = threadIdx.x >> 6;
    for (int c0 = 0; c0 < a.dc; c0 += 128) {
        const int c = c0 + threadIdx.x;
        float acc = 0.f;
        for (long base = s0; base < s1; base += 128) {
            const long n_mine = base + threadIdx.x;
            const unsigned long long b = __ballot(n_mine < s1 && ids[n_mine] == row);
            __syncthreads();                                            // the previous window's masks are consumed
            if ((threadIdx.x & 63) == 0) hit[wave] = b;
            __syncthreads();
            if (c < a.dc)
                for (int w = 0; w < 2; ++w)
                    for (unsigned long long m = hit[w]; m != 0; m &= m - 1) {
                        const long n = base + 64 * w + __builtin_ctzll(m);
                        float g = a.dout[n * F + col0 + c];
                        if (a.drop.thresh != 0u)
                            g *= dropout_scale1(a.drop.seed, 3u, (uint64_t)(n * F + col0 + c), a.drop.thresh, a.drop.inv_keep);
                        acc += g;
                    }
        }
        if (c < a.dc) dst[c] = acc;
    }
}

__global__ __launch_bounds__(256) void features_tables_reduce_kernel(FeatArgs a, int n_chunks, const float* partial) {
    const int rows = (a.n_cat - 1) + (a.n_sub - 1);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * a.dc) return;
    const int b = (int)(i / a.dc), c = (int)(i - (long)b * a.dc);
    float acc = 0.f;
    for (int k = 0; k < n_chunks; ++k) acc += partial[((long)k * rows + b) * a.dc + c];
    const bool is_cat = b < a.n_cat - 1;
    const int row = is_cat ? b + 1 : b - (a.n_cat - 1) + 1;
    (is_cat ? a.d_cat_table : a.d_sub_table)[(long)row * a.dc + c] += acc;
}

int launch_features_bwd(const FeatArgs& a, hipStream_t stream) {
    if (a.n <= 0) return NRMS_OK;
    TimingScope ts("features_bwd", stream);
    const int rows = (a.n_cat - 1) + (a.n_sub - 1);
    // chunked table sums: the partial sums live in d_title (n x d_text floats, overwritten by the text kernel AFTERWARDS)
    long chunks = rows > 0 ? 8192 / rows : 0;
    chunks = chunks > 64 ? 64 : chunks;
    const long windows = (a.n + 127) / 128;
    if (chunks > windows) chunks = windows;
    while (chunks > 1 && chunks * rows * (long)a.dc > a.n * (long)a.dt) --chunks;
    if (rows > 0 && chunks > 1) {
        const long chunk = ((a.n + chunks - 1) / chunks + 127) / 128 * 128;
        const int n_chunks = (int)((a.n + chunk - 1) / chunk);
        hipLaunchKernelGGL(features_bwd_tables_chunk_kernel, dim3(rows, n_chunks), dim3(128), 0, stream, a, chunk, a.d_title);
        hipLaunchKernelGGL(features_tables_reduce_kernel, dim3(cdiv((long)rows * a.dc, 256)), dim3(256), 0, stream, a, n_chunks, (const float*)a.d_title);
    }
    hipLaunchKernelGGL(features_bwd_text_kernel, dim3((int)(a.n > 16384 ? 16384 : a.n)), dim3(256), 0, stream, a);
    if (rows > 0 && chunks <= 1) hipLaunchKernelGGL(features_bwd_tables_kernel, dim3(rows), dim3(128), 0, stream, a);
    return check_launch("features_bwd");
}

}  // namespace nrms
