// Additive-attention pooling (AdditiveAttention, /root/reference/MIND_2020/model/nrms_v0.py:100-126),
// click scores (:205-216,:272-274), cross-entropy with label 0 (train_eval.py:63,116-117),
// fused Adam (train_eval.py:48,127) and the dropout-mask export used by the parity tests.
#include "gemm.h"

namespace nrms {

// =======================================================================================
// Forward: one workgroup owns `spb` whole sequences (rows_per_tile = spb*S <= 128):
//   T = tanh(C Wa^T + ba) on the MFMA, s = T.q reduced across the tile's columns in-register,
//   w = softmax_S(s), out = sum_s w_s C_s.  T and w are saved for the backward.
// =======================================================================================
template <int NT>
__global__ __launch_bounds__(256, 2) void addattn_fwd_kernel(AddFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (NT_BM + 16 * NT) * NT_BK];
    __shared__ float sc[NT_BM];

    const NTArgs& g = a.g;
    const int row0 = blockIdx.x * g.rows_per_tile;
    const int rows_valid = min(g.rows_per_tile, g.M - row0);
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_nt_mainloop<NT, A_PLAIN>(g, row0, rows_valid, 0, acc, lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    // this lane's columns n = 16 nt + r16: bias and query-vector slices once per tile (not per element)
    float bcol[NT], qcol[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * nt + r16;
        const bool ok = n < g.N;
        bcol[nt] = ok ? g.bias[n] : 0.f;
        qcol[nt] = ok ? a.qv[n] : 0.f;              // q = 0 also zeroes the padded columns' contribution
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rl = 32 * wave + 16 * mt + 4 * kq + reg;
            float part = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float t = tanhf(acc[mt][nt][reg] + bcol[nt]);     // exact-parity mode keeps libm's tanhf
                part += t * qcol[nt];
                acc[mt][nt][reg] = t;
            }
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            part += __shfl_xor(part, 8, 64);
            if (r16 == 0) sc[rl] = part;
        }
    if (a.T != nullptr) {
        // T [M, q] in whole rows through the per-wave LDS strips (the stage buffers are dead now)
        static_assert(4 * 8 * (16 * NT + 8) <= 2 * (NT_BM + 16 * NT) * NT_BK, "epilogue strips must fit the stage buffers");
        __syncthreads();
        NTArgs tg = g;
        tg.bias = nullptr; tg.C = a.T; tg.ldc = g.N;
        nt_epilogue<NT, E_STORE>(tg, acc, row0, rows_valid, 0, wave, lane, lds + wave * 8 * (16 * NT + 8));
    }
    __syncthreads();
    // softmax over each sequence: one wave per sequence, a lane per position (see addattn_fwd_bf16_kernel)
    const int spb = rows_valid / a.S;
    for (int sq = wave; sq < spb; sq += 4) {
        float* sp = sc + sq * a.S;
        float v = -1e30f;
        if (lane < a.S) {
            v = sp[lane];
            if (a.mask != nullptr && a.mask[(long)row0 + sq * a.S + lane] == 0) v = -1e9f;
        }
        float mx = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float e = lane < a.S ? expf(v - mx) : 0.f;
        float sum = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float w = e / sum;
        if (lane < a.S) {
            sp[lane] = w;
            if (a.wout != nullptr) a.wout[(long)row0 + sq * a.S + lane] = w;
        }
    }
    __syncthreads();
    const long seq0 = row0 / a.S;
    for (int idx = tid; idx < spb * a.d; idx += 256) {
        const int sq = idx / a.d, c = idx - sq * a.d;
        const float* crow = g.A + ((long)row0 + sq * a.S) * g.lda + c;
        const float* w = sc + sq * a.S;
        float o = 0.f;
        for (int i = 0; i < a.S; ++i) o += w[i] * crow[(long)i * g.lda];
        a.out[(seq0 + sq) * a.d + c] = o;
    }
}

template <int NT>
static int launch_addfwd_inst(const AddFwdArgs& a, hipStream_t stream) {
    TimingScope ts("addattn_fwd", stream);
    hipLaunchKernelGGL((addattn_fwd_kernel<NT>), dim3(cdiv(a.g.M, a.g.rows_per_tile)), dim3(256), 0, stream, a);
    return check_launch("addattn_fwd");
}

int launch_addattn_fwd(int n_seq, int S, int d, int q, const float* ctx, const float* w_add, const float* b_add,
                       const float* q_vec, float* T, float* wout, float* out, const uint8_t* mask, int npass,
                       void* wplanes, hipStream_t stream) {
    if (n_seq <= 0) return NRMS_OK;
    AddFwdArgs a{};
    a.g.M = n_seq * S; a.g.N = q; a.g.K = d;
    a.g.rows_per_tile = (NT_BM / S) * S;
    a.g.A = ctx; a.g.lda = d; a.g.W = w_add; a.g.bias = b_add;
    a.qv = q_vec; a.T = T; a.wout = wout; a.out = out; a.S = S; a.d = d; a.mask = mask;
    if (npass != 0 && q <= 208) return launch_addattn_fwd_bf16(npass, a, wplanes, stream);
    if (q <= 32) return launch_addfwd_inst<2>(a, stream);
    if (q <= 64) return launch_addfwd_inst<4>(a, stream);
    if (q <= 128) return launch_addfwd_inst<8>(a, stream);
    if (q <= 208) return launch_addfwd_inst<13>(a, stream);
    if (q <= 256) return launch_addfwd_inst<16>(a, stream);
    set_error("addattn_fwd: q_dim=%d > 256 unsupported", q);
    return NRMS_EINVAL;
}

// =======================================================================================
// Backward, row part: one wave per sequence.
//   dw_s = C_s . dout ; ds_s = w_s (dw_s - sum_t w_t dw_t) ; dq[n] += sum_s ds_s T[s][n]
// ds feeds the dZ loaders of the two GEMMs; dq goes to a per-wave partial row (deterministic
// two-stage sum).
// =======================================================================================
constexpr int ROWS_WPB = 4;

__global__ __launch_bounds__(64 * ROWS_WPB) void addattn_bwd_rows_kernel(
    int n_seq, int S, int d, int q, const float* ctx, const float* dout, const float* w, const float* T, float* ds,
    float* dq_partial, const uint8_t* mask) {
    // Four 16-lane groups take one row each: a row dot is ceil(d/64) independent float4 loads per lane
    // and a 4-step xor reduction, so a whole sequence is ~S/4 * ceil(d/64) loads in flight per lane
    // instead of S dependent (load, 6-step reduce) rounds.
    __shared__ float dwl[ROWS_WPB][64], dsl[ROWS_WPB][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, grp = lane >> 4;
    const long gw = (long)blockIdx.x * ROWS_WPB + wave, nw = (long)gridDim.x * ROWS_WPB;
    constexpr int MAXC = 8;                 // float4 chunks per lane: d <= 16 * 4 * MAXC = 512
    const int d4 = d >> 2, nch = (d4 + 15) >> 4;
    const bool qok = 4 * lane < q;
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
    for (long seq = gw; seq < n_seq; seq += nw) {
        const float* c = ctx + seq * S * d;
        const float* g = dout + seq * d;
        f32x4 gv[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c4 = l16 + 16 * i;
            gv[i] = (i < nch && c4 < d4) ? *reinterpret_cast<const f32x4*>(g + 4 * min(c4, d4 - 1)) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(i < nch && c4 < d4)) gv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int r0 = 0; r0 < S; r0 += 4) {
            const int r = min(r0 + grp, S - 1);
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                if (i < nch) {
                    const int c4 = min(l16 + 16 * i, d4 - 1);
                    const f32x4 cv = *reinterpret_cast<const f32x4*>(c + (long)r * d + 4 * c4);
                    p += cv[0] * gv[i][0] + cv[1] * gv[i][1] + cv[2] * gv[i][2] + cv[3] * gv[i][3];
                }
            }
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            p += __shfl_xor(p, 4, 64);
            p += __shfl_xor(p, 8, 64);
            if (l16 == 0 && r0 + grp < S) dwl[wave][r0 + grp] = p;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float my_dw = lane < S ? dwl[wave][lane] : 0.f;
        const float my_w = lane < S ? w[seq * S + lane] : 0.f;
        const float dot = wave_sum(my_w * my_dw);
        float my_ds = my_w * (my_dw - dot);
        // masked_fill passes no gradient to the overwritten scores
        if (mask != nullptr && lane < S && mask[seq * S + lane] == 0) my_ds = 0.f;
        if (lane < S) ds[seq * S + lane] = my_ds;
        dsl[wave][lane] = my_ds;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float* t = T + seq * S * q + 4 * (qok ? lane : 0);
        for (int s2 = 0; s2 < S; ++s2) {
            const f32x4 tv = *reinterpret_cast<const f32x4*>(t + (long)s2 * q);
            dq += dsl[wave][s2] * tv;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (qok) *reinterpret_cast<f32x4*>(dq_partial + gw * q + 4 * lane) = dq;
}

__global__ __launch_bounds__(1024) void colsum_add_kernel(const float* partial, int rows, int cols, float* out) {
    // one block per 16 columns: 64 row-groups x 16 columns, 4 independent loads per thread in flight (the column
    // count is small -- q = 200 gives 13 blocks -- so the kernel lives off memory-level parallelism per CU)
    __shared__ float red[64][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int r = rg;
        for (; r + 192 < rows; r += 256) {
            s0 += partial[(long)r * cols + c];
            s1 += partial[(long)(r + 64) * cols + c];
            s2 += partial[(long)(r + 128) * cols + c];
            s3 += partial[(long)(r + 192) * cols + c];
        }
        for (; r < rows; r += 64) s0 += partial[(long)r * cols + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) t += red[i][cl];
        out[c] += t;
    }
}

int addattn_bwd_rows_waves(int n_seq) {
    int blocks = cdiv(n_seq, ROWS_WPB);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return blocks * ROWS_WPB;
}

int launch_addattn_bwd_rows(int n_seq, int S, int d, int q, const float* ctx, const float* dout, const float* w,
                            const float* T, float* ds, float* dq_partial, float* dq, const uint8_t* mask,
                            hipStream_t stream) {
    if (n_seq <= 0) return NRMS_OK;
    const int waves = addattn_bwd_rows_waves(n_seq);
    {
        TimingScope ts("addattn_bwd_rows", stream);
        hipLaunchKernelGGL(addattn_bwd_rows_kernel, dim3(waves / ROWS_WPB), dim3(64 * ROWS_WPB), 0, stream, n_seq, S,
                           d, q, ctx, dout, w, T, ds, dq_partial, mask);
        int rc = check_launch("addattn_bwd_rows");
        if (rc) return rc;
    }
    return launch_colsum_add(dq_partial, waves, q, dq, stream);
}

// out[c] += sum_r partial[r][c], rows summed in a fixed order
int launch_colsum_add(const float* partial, int rows, int cols, float* out, hipStream_t stream) {
    TimingScope ts("colsum_add", stream);
    hipLaunchKernelGGL(colsum_add_kernel, dim3(cdiv(cols, 16)), dim3(1024), 0, stream, partial, rows, cols, out);
    return check_launch("colsum_add");
}

// =======================================================================================
// Click scores, CE loss, Adam, dropout mask export
// =======================================================================================
__global__ void click_fwd_kernel(int B, int C, int d, const float* cand, const float* user, const uint8_t* mask,
                                 float* scores) {
    const int lane = threadIdx.x & 63;
    const long u = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (u >= (long)B * C) return;
    const long b = u / C;
    const float* cv = cand + u * d;
    const float* uv = user + b * d;
    float p = 0.f;
    for (int k = lane; k < d; k += 64) p += cv[k] * uv[k];
    p = wave_sum(p);
    if (lane == 0) scores[u] = (mask != nullptr && mask[u] == 0) ? -1e9f : p;
}

// candidate vectors by index into a news-vector table; an index outside [0, n_vec) reads row 0 and scores NaN
__global__ void click_indexed_kernel(int B, int C, int d, const float* vec, long n_vec, const int* index, const float* user,
                                     const uint8_t* mask, float* scores) {
    const int lane = threadIdx.x & 63;
    const long u = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (u >= (long)B * C) return;
    const long b = u / C;
    const long r = index[u];
    const bool ok = r >= 0 && r < n_vec;
    const float* cv = vec + (ok ? r : 0) * d;
    const float* uv = user + b * d;
    float p = 0.f;
    for (int k = lane; k < d; k += 64) p += cv[k] * uv[k];
    p = wave_sum(p);
    if (lane == 0) scores[u] = (mask != nullptr && mask[u] == 0) ? -1e9f : (ok ? p : __builtin_nanf(""));
}

__global__ void click_bwd_kernel(int B, int C, int d, const float* cand, const float* user, const uint8_t* mask,
                                 const float* dscores, float* dcand, float* duser) {
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        const float uv = user[(long)b * d + k];
        float du = 0.f;
        for (int c = 0; c < C; ++c) {
            const long u = (long)b * C + c;
            const float g = (mask != nullptr && mask[u] == 0) ? 0.f : dscores[u];
            dcand[u * d + k] = g * uv;
            du += g * cand[u * d + k];
        }
        duser[(long)b * d + k] = du;
    }
}

__global__ void ce_loss_kernel(int B, int C, const float* scores, float* loss_sum, float* dscores, float gscale) {
    __shared__ float red[256];
    float local = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* s = scores + (long)b * C;
        float mx = -3.0e38f;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, s[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(s[c] - mx);
        const float lse = mx + logf(sum);
        local += lse - s[0];
        if (dscores != nullptr) {
            const float inv = 1.0f / sum;
            for (int c = 0; c < C; ++c) {
                const float p = expf(s[c] - mx) * inv;
                dscores[(long)b * C + c] = (p - (c == 0 ? 1.0f : 0.0f)) * gscale;
            }
        }
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_sum[0] += red[0];
}

// GUARD: an element whose gradient is not finite (an fp16 overflow of the fused backward, an inf / nan loss) is left out of
// the update -- p, m and v keep their values, so Adam's moments cannot be poisoned for the rest of training -- and counted into
// *n_bad (one atomic per wavefront that saw any; none in a healthy step).  Every rank of a data-parallel job sees the same
// reduced gradient, hence skips the same elements: replicas stay identical.
template <bool GUARD>
__global__ void adam_kernel(size_t n4, size_t n, float* p, const float* g, float* m, float* v, float step_size,
                            float b1, float b2, float omb1, float omb2, float inv_sqrt_bc2, float eps, float gscale, int* n_bad) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * gscale;
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (GUARD && (__float_as_uint(gv[e]) & 0x7F800000u) == 0x7F800000u) { ++bad; continue; }
            mv[e] = b1 * mv[e] + omb1 * gv[e];
            vv[e] = b2 * vv[e] + omb2 * gv[e] * gv[e];
            const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
            pv[e] -= step_size * (mv[e] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    const size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float gv = g[t] * gscale;
        if (GUARD && (__float_as_uint(gv) & 0x7F800000u) == 0x7F800000u) {
            ++bad;
        } else {
            const float mv = b1 * m[t] + omb1 * gv;
            const float vv = b2 * v[t] + omb2 * gv * gv;
            m[t] = mv; v[t] = vv;
            p[t] -= step_size * (mv / (sqrtf(vv) * inv_sqrt_bc2 + eps));
        }
    }
    if (GUARD) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
        if (bad != 0 && (threadIdx.x & 63) == 0) atomicAdd(n_bad, bad);
    }
}

// nrms_grad_guard: non-finite elements -> 0, counted
__global__ void grad_guard_kernel(size_t n, float* g, int* n_bad) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int bad = 0;
    const size_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 gv = reinterpret_cast<f32x4*>(g)[i];
        int b = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((__float_as_uint(gv[e]) & 0x7F800000u) == 0x7F800000u) { gv[e] = 0.f; ++b; }
        if (b) { reinterpret_cast<f32x4*>(g)[i] = gv; bad += b; }
    }
    for (size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        if ((__float_as_uint(g[t]) & 0x7F800000u) == 0x7F800000u) { g[t] = 0.f; ++bad; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if (bad != 0 && (threadIdx.x & 63) == 0) atomicAdd(n_bad, bad);
}

__global__ void keep_mask_kernel(uint64_t seed, uint32_t site, long groups, uint32_t thresh, uint8_t* keep) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups) return;
    uint32_t r[4];
    philox4x32_7(seed, (uint64_t)i, site, r);
    uchar4 k;
    k.x = r[0] >= thresh; k.y = r[1] >= thresh; k.z = r[2] >= thresh; k.w = r[3] >= thresh;
    reinterpret_cast<uchar4*>(keep)[i] = k;
}

// 16-bit-field scheme (dropout_scale8): element i of the flat layout takes field i & 7 of call i >> 3
__global__ void keep_mask16_kernel(uint64_t seed, uint32_t site, long groups8, uint32_t thresh16, uint8_t* keep) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups8) return;
    uint32_t r[4];
    philox4x32_7(seed, (uint64_t)i, site, r);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        keep[8 * i + 2 * w] = (r[w] & 0xFFFFu) >= thresh16;
        keep[8 * i + 2 * w + 1] = (r[w] >> 16) >= thresh16;
    }
}

// Per-impression ROC-AUC on the un-padded prefix (train_eval.py:219-227 `auc_score(y_true,
// rank_score[i][:len(y_true)])`, evaluation.py:26-27 = sklearn roc_auc_score): the Mann-Whitney
// statistic  (#{pos > neg} + 0.5 #{pos == neg}) / (n_pos n_neg)  counted exactly in integers and
// divided once in float64, so it equals the reference to the last bit.  One wave per impression.
__global__ __launch_bounds__(256) void impression_auc_kernel(int n_imp, int max_c, const float* scores,
                                                             const uint8_t* labels, const int32_t* lens, double* auc) {
    const int lane = threadIdx.x & 63;
    const int imp = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (imp >= n_imp) return;
    const int n = min(lens[imp], max_c);
    const float* s = scores + (long)imp * max_c;
    const uint8_t* y = labels + (long)imp * max_c;
    unsigned long long twice = 0;      // 2*greater + equal, over (pos, neg) pairs
    unsigned npos = 0;
    for (int i = lane; i < n; i += 64) {
        if (y[i] == 0) continue;
        ++npos;
        const float si = s[i];
        for (int j = 0; j < n; ++j) {
            if (y[j] != 0) continue;
            const float sj = s[j];
            twice += si > sj ? 2u : (si == sj ? 1u : 0u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        twice += __shfl_xor(twice, o, 64);
        npos += __shfl_xor(npos, o, 64);
    }
    if (lane == 0) {
        const unsigned nneg = (unsigned)n - npos;
        auc[imp] = (npos == 0 || nneg == 0) ? __longlong_as_double(0x7FF8000000000000LL)      // undefined: NaN
                                             : 0.5 * (double)twice / ((double)npos * (double)nneg);
    }
}

}  // namespace nrms

using namespace nrms;

extern "C" int nrms_impression_auc(int32_t n_imp, int32_t max_c, const float* scores, const uint8_t* labels,
                                   const int32_t* lens, double* auc, void* stream) {
    NRMS_REQUIRE(n_imp >= 0 && max_c > 0 && scores && labels && lens && auc, "impression_auc: bad arguments");
    if (n_imp == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("impression_auc", s);
    hipLaunchKernelGGL(impression_auc_kernel, dim3(cdiv(n_imp, 4)), dim3(256), 0, s, n_imp, max_c, scores, labels, lens, auc);
    return check_launch("impression_auc");
}

extern "C" int nrms_click_score_fwd(int32_t B, int32_t C, int32_t d, const float* cand, const float* user,
                                    const uint8_t* mask, float* scores, void* stream) {
    NRMS_REQUIRE(B >= 0 && C > 0 && d > 0 && cand && user && scores, "click_score_fwd: bad arguments");
    if (B == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("click_fwd", s);
    hipLaunchKernelGGL(click_fwd_kernel, dim3(cdiv((long)B * C, 4)), dim3(256), 0, s, B, C, d, cand, user, mask, scores);
    return check_launch("click_fwd");
}

extern "C" int nrms_click_score_indexed(int32_t B, int32_t C, int32_t d, const float* news_vec, int64_t n_vec, const int32_t* index,
                                        const float* user, const uint8_t* mask, float* scores, void* stream) {
    NRMS_REQUIRE(B >= 0 && C > 0 && d > 0 && n_vec > 0 && news_vec && index && user && scores, "click_score_indexed: bad arguments");
    if (B == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("click_fwd", s);
    hipLaunchKernelGGL(click_indexed_kernel, dim3(cdiv((long)B * C, 4)), dim3(256), 0, s, B, C, d, news_vec, (long)n_vec, index,
                       user, mask, scores);
    return check_launch("click_indexed");
}

extern "C" int nrms_click_score_bwd(int32_t B, int32_t C, int32_t d, const float* cand, const float* user,
                                    const uint8_t* mask, const float* dscores, float* dcand, float* duser,
                                    void* stream) {
    NRMS_REQUIRE(B >= 0 && C > 0 && d > 0 && cand && user && dscores && dcand && duser, "click_score_bwd: bad arguments");
    if (B == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("click_bwd", s);
    hipLaunchKernelGGL(click_bwd_kernel, dim3(B), dim3(128), 0, s, B, C, d, cand, user, mask, dscores, dcand, duser);
    return check_launch("click_bwd");
}

extern "C" int nrms_ce_loss_fwd_bwd(int32_t B, int32_t C, const float* scores, float* loss_sum, float* dscores,
                                    float grad_scale, void* stream) {
    NRMS_REQUIRE(B >= 0 && C > 0 && scores && loss_sum, "ce_loss: bad arguments");
    if (B == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("ce_loss", s);
    hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, s, B, C, scores, loss_sum, dscores, grad_scale);
    return check_launch("ce_loss");
}

static int adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                     double beta2, double eps, int32_t step, float grad_scale, int32_t* n_nonfinite, void* stream) {
    NRMS_REQUIRE(param && grad && exp_avg && exp_avg_sq && step >= 1, "adam_step: bad arguments");
    NRMS_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                 "adam_step: buffers must be 16-byte aligned");
    if (n == 0) return NRMS_OK;
    // torch.optim.Adam: step_size = lr / (1 - b1^t); denom = sqrt(v)/sqrt(1 - b2^t) + eps
    // hyper-parameters arrive as doubles, as torch holds them: 1 - beta is formed in double and rounded once
    // (float(1 - 0.999) != 1 - float(0.999): 1.3e-5 relative in exp_avg_sq)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    TimingScope ts("adam", s);
    if (n_nonfinite != nullptr)
        hipLaunchKernelGGL(adam_kernel<true>, dim3(blocks), dim3(256), 0, s, n4, n, param, grad, exp_avg, exp_avg_sq, step_size,
                           (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), inv_sqrt_bc2, (float)eps, grad_scale,
                           (int*)n_nonfinite);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, s, n4, n, param, grad, exp_avg, exp_avg_sq, step_size,
                           (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), inv_sqrt_bc2, (float)eps, grad_scale,
                           (int*)nullptr);
    return check_launch("adam");
}

extern "C" int nrms_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr,
                              double beta1, double beta2, double eps, int32_t step, float grad_scale, void* stream) {
    return adam_step(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale, nullptr, stream);
}

extern "C" int nrms_adam_step_guarded(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr,
                                      double beta1, double beta2, double eps, int32_t step, float grad_scale,
                                      int32_t* n_nonfinite, void* stream) {
    NRMS_REQUIRE(n_nonfinite != nullptr, "adam_step_guarded: null counter");
    return adam_step(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale, n_nonfinite, stream);
}

extern "C" int nrms_grad_guard(size_t n, float* grad, int32_t* n_nonfinite, void* stream) {
    NRMS_REQUIRE(n == 0 || (grad && n_nonfinite), "grad_guard: null argument");
    if (n == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    TimingScope ts("grad_guard", s);
    hipLaunchKernelGGL(grad_guard_kernel, dim3((int)blocks), dim3(256), 0, s, n, grad, (int*)n_nonfinite);
    return check_launch("grad_guard");
}

extern "C" int nrms_dropout_keep_mask(uint64_t seed, int32_t site, int64_t n_rows, int32_t d, float p_drop,
                                      uint8_t* keep, void* stream) {
    NRMS_REQUIRE(keep && n_rows >= 0 && d > 0 && (d & 3) == 0 && site >= 0, "dropout_keep_mask: bad arguments");
    const long groups = (long)n_rows * d / 4;
    if (groups == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (site & NRMS_DROPOUT_FIELDS16) {
        NRMS_REQUIRE((d & 7) == 0, "dropout_keep_mask: the 16-bit-field scheme needs d %% 8 == 0");
        hipLaunchKernelGGL(keep_mask16_kernel, dim3(cdiv(groups / 2, 256)), dim3(256), 0, s, seed, (uint32_t)(site & 0xFF), groups / 2,
                           drop_threshold16(p_drop), keep);
        return check_launch("keep_mask16");
    }
    hipLaunchKernelGGL(keep_mask_kernel, dim3(cdiv(groups, 256)), dim3(256), 0, s, seed, (uint32_t)site, groups,
                       drop_threshold(p_drop), keep);
    return check_launch("keep_mask");
}
