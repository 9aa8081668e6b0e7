// All-padding sequences of the output-projection topology (nrms_naml's word-level encoder, model/nrms_naml.py:42-100,121-177),
// in closed form: nrms_encoder_empty_fwd / _bwd (include/nrms_hip.h).
//
// A sequence whose word ids are all 0 feeds S copies of the (zero) padding row into the encoder: every Q, K, V row is the bias,
// the attention probabilities are 1 / S before their dropout, so the head-h block of attention row i is c_ih b_v^(h) with
//     c_ih = (kept keys of query i in head h) / (S (1 - p))                       (1 without dropout)
// -- h scalars per row.  Everything after it is a function of those scalars and of h + 1 vectors that depend on the weights only:
//     u_h = W_O[:, head h] b_v^(h)      y_i = sum_h c_ih u_h + b_O                  (output_linear, :61-75)
//     v_h = W_add u_h, v_0 = W_add b_O + b_add      t_i = tanh(sum_h c_ih v_h + v_0)   (AdditiveAttention, :77-100)
//     s_i = <t_i, q_vec>, w = softmax_i(s), out = sum_h (sum_i w_i c_ih) u_h + b_O
// and the backward needs, over ALL such sequences, only  A_h = sum c_ih dZ_i (h + 1 vectors of q),  sum ds_i t_i,
// G_h = sum (sum_i w_i c_ih) g (h + 1 vectors of d):
//     D_h = G_h + W_add^T A_h;  d(b_O) += D_0;  d(b_add) += A_0;  d(q_vec) += sum ds_i t_i;
//     d(W_add) += sum_h A_h u_h^T + A_0 b_O^T;  d(W_O)[:, head h] += D_h b_v^(h)^T;  d(b_v^(h)) += W_O[:, head h]^T D_h
// (d(W_qkv) = 0: the inputs are zero rows; dQ = 0 and the dK rows sum to zero; the padding row takes no gradient).  41 % of the
// sequences of a MIND-shaped nrms_naml batch are such history-padding slots; the GEMM chain spent 4.3 ms of its 26.5 ms step on their
// rows.  The caller runs the chain on the other sequences (compacted, with desc.seq_index so that their dropout counters stay those
// of the full batch) and this file on the all-padding ones.  The dropout decisions are the attention kernel's own (site 2, element
// ((seq * h + head) * S + i) * S + j of the FULL batch's numbering): nrms_dropout_keep_mask replays them for the oracle.
// One wave per sequence, fp32 FMA arithmetic; partial sums per wave, added in a fixed order.
#include "common.h"

namespace nrms {

constexpr int EM_HMAX = 8;          // heads
constexpr int EM_QL = 4;            // q <= 256: columns lane + 64 j
constexpr int EM_DL = 8;            // d <= 512
constexpr int EM_WPB = 4;

struct EmptyArgs {
    int n_seq, S, d, h, dk, q;
    const int* seq_index;           // [n_seq] sequence numbers in the full batch (dropout counters) or null (0 .. n_seq - 1)
    Dropout pd;                     // dropout on the attention probabilities
    const float* consts;            // u [h][d] | v [h + 1][q]  (v[0] = v_0)
    const float* b_o;
    const float* q_vec;
    float* out;                     // forward  [n_seq][d]
    const float* dout;              // backward [n_seq][d]
    float* partial;                 // backward [n_waves][(h + 1) q + q + (h + 1) d]
};

// consts: u_h = W_O[:, head h] b_v^(h);  v_h = W_add u_h (h = 1 .. H), v_0 = W_add b_O + b_add.  One workgroup.
__global__ __launch_bounds__(1024) void empty_consts_kernel(int d, int h, int dk, int q, const float* b_v, const float* w_o, const float* b_o,
                                                            const float* w_add, const float* b_add, float* consts) {
    float* u = consts;
    float* v = consts + (long)h * d;
    for (int i = threadIdx.x; i < h * d; i += 1024) {
        const int hh = i / d, o = i - hh * d;
        float acc = 0.f;
        for (int j = 0; j < dk; ++j) acc += w_o[(long)o * d + hh * dk + j] * b_v[hh * dk + j];
        u[i] = acc;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = threadIdx.x; i < (h + 1) * q; i += 1024) {
        const int hh = i / q, n = i - hh * q;
        const float* x = hh == 0 ? b_o : u + (long)(hh - 1) * d;
        float acc = hh == 0 ? b_add[n] : 0.f;
        for (int o = 0; o < d; ++o) acc += w_add[(long)n * d + o] * x[o];
        v[i] = acc;
    }
}

// c_ih of this lane's row: the kept keys of query `row` in unit (seq, head) -- the attention kernel's counters (attention.hip,
// prob_keep_bits_col)
__device__ __forceinline__ float empty_keep_factor(const Dropout& pd, long unit, int S, int row) {
    if (pd.thresh == 0u) return 1.0f;
    const uint64_t e0 = (uint64_t)((unit * S + row) * S), e1 = e0 + S;
    int cnt = 0;
    for (uint64_t g = e0 >> 2; g <= (e1 - 1) >> 2; ++g) {
        uint32_t r[4];
        philox4x32_7(pd.seed, g, 2u, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t e = 4 * g + k;
            cnt += (e >= e0 && e < e1 && r[k] >= pd.thresh) ? 1 : 0;
        }
    }
    return (float)cnt * pd.inv_keep / (float)S;
}

// shared by both directions: c (LDS [64][EM_HMAX], also returned for this lane's row), s_i -> w_i of this lane's row
template <bool BWD>
__device__ __forceinline__ void empty_rows(const EmptyArgs& a, long seq, int lane, float (*cs)[EM_HMAX], const float (&vreg)[EM_HMAX + 1][EM_QL],
                                           const float (&qv)[EM_QL], float (&c)[EM_HMAX], float& w_i) {
    const long sidx = a.seq_index != nullptr ? (long)a.seq_index[seq] : seq;
#pragma unroll
    for (int hh = 0; hh < EM_HMAX; ++hh) {
        c[hh] = (hh < a.h && lane < a.S) ? empty_keep_factor(a.pd, sidx * a.h + hh, a.S, lane) : 0.f;
        cs[lane][hh] = c[hh];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float s_mine = -3.0e38f;
    for (int i = 0; i < a.S; ++i) {
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < EM_QL; ++j) {
            if (lane + 64 * j < a.q) {
                float z = vreg[0][j];
#pragma unroll
                for (int hh = 0; hh < EM_HMAX; ++hh)
                    if (hh < a.h) z += cs[i][hh] * vreg[hh + 1][j];
                part += tanhf(z) * qv[j];
            }
        }
        part = wave_sum(part);
        if (lane == i) s_mine = part;
    }
    const float m = wave_max(s_mine);
    const float e = lane < a.S ? __expf(s_mine - m) : 0.f;
    w_i = e / wave_sum(e);
}

__device__ __forceinline__ void empty_load_consts(const EmptyArgs& a, int lane, float (&vreg)[EM_HMAX + 1][EM_QL], float (&qv)[EM_QL]) {
    const float* v = a.consts + (long)a.h * a.d;
#pragma unroll
    for (int j = 0; j < EM_QL; ++j) {
        const int n = lane + 64 * j;
        qv[j] = n < a.q ? a.q_vec[n] : 0.f;
#pragma unroll
        for (int hh = 0; hh <= EM_HMAX; ++hh) vreg[hh][j] = (n < a.q && hh <= a.h) ? v[(long)hh * a.q + n] : 0.f;
    }
}

__global__ __launch_bounds__(64 * EM_WPB) void empty_fwd_kernel(EmptyArgs a) {
    __shared__ float cs_all[EM_WPB][64][EM_HMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*cs)[EM_HMAX] = cs_all[wave];
    float vreg[EM_HMAX + 1][EM_QL], qv[EM_QL];
    empty_load_consts(a, lane, vreg, qv);
    for (long seq = (long)blockIdx.x * EM_WPB + wave; seq < a.n_seq; seq += (long)gridDim.x * EM_WPB) {
        float c[EM_HMAX], w_i;
        empty_rows<false>(a, seq, lane, cs, vreg, qv, c, w_i);
        float wc[EM_HMAX];
#pragma unroll
        for (int hh = 0; hh < EM_HMAX; ++hh) wc[hh] = hh < a.h ? wave_sum(w_i * c[hh]) : 0.f;
#pragma unroll
        for (int j = 0; j < EM_DL; ++j) {
            const int o = lane + 64 * j;
            if (o < a.d) {
                float y = a.b_o[o];
#pragma unroll
                for (int hh = 0; hh < EM_HMAX; ++hh)
                    if (hh < a.h) y += wc[hh] * a.consts[(long)hh * a.d + o];
                a.out[seq * a.d + o] = y;
            }
        }
        __builtin_amdgcn_wave_barrier();                                 // cs is rewritten by the next sequence
    }
}

__global__ __launch_bounds__(64 * EM_WPB) void empty_bwd_kernel(EmptyArgs a) {
    __shared__ float cs_all[EM_WPB][64][EM_HMAX];
    __shared__ float ds_all[EM_WPB][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*cs)[EM_HMAX] = cs_all[wave];
    float* dsb = ds_all[wave];
    float vreg[EM_HMAX + 1][EM_QL], qv[EM_QL];
    empty_load_consts(a, lane, vreg, qv);
    float A[EM_HMAX + 1][EM_QL], dqv[EM_QL], G[EM_HMAX + 1][EM_DL];
#pragma unroll
    for (int hh = 0; hh <= EM_HMAX; ++hh) {
#pragma unroll
        for (int j = 0; j < EM_QL; ++j) A[hh][j] = 0.f;
#pragma unroll
        for (int j = 0; j < EM_DL; ++j) G[hh][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < EM_QL; ++j) dqv[j] = 0.f;
    for (long seq = (long)blockIdx.x * EM_WPB + wave; seq < a.n_seq; seq += (long)gridDim.x * EM_WPB) {
        float c[EM_HMAX], w_i;
        empty_rows<true>(a, seq, lane, cs, vreg, qv, c, w_i);
        // g = d(out) of this sequence; <g, u_h>, <g, b_O>
        float g[EM_DL], gu[EM_HMAX + 1];
#pragma unroll
        for (int hh = 0; hh <= EM_HMAX; ++hh) gu[hh] = 0.f;
#pragma unroll
        for (int j = 0; j < EM_DL; ++j) {
            const int o = lane + 64 * j;
            g[j] = o < a.d ? a.dout[seq * a.d + o] : 0.f;
            if (o < a.d) {
                gu[0] += g[j] * a.b_o[o];
#pragma unroll
                for (int hh = 0; hh < EM_HMAX; ++hh)
                    if (hh < a.h) gu[hh + 1] += g[j] * a.consts[(long)hh * a.d + o];
            }
        }
#pragma unroll
        for (int hh = 0; hh <= EM_HMAX; ++hh) gu[hh] = hh <= a.h ? wave_sum(gu[hh]) : 0.f;
        // softmax backward over the rows (lane = row)
        float dw = gu[0];
#pragma unroll
        for (int hh = 0; hh < EM_HMAX; ++hh) dw += c[hh] * gu[hh + 1];
        const float sumwd = wave_sum(w_i * dw);
        dsb[lane] = w_i * (dw - sumwd);                                  // 0 beyond the sequence (w_i = 0)
        // G_h += (sum_i w_i c_ih) g, G_0 += g
#pragma unroll
        for (int hh = 0; hh < EM_HMAX; ++hh) {
            const float wc = hh < a.h ? wave_sum(w_i * c[hh]) : 0.f;
#pragma unroll
            for (int j = 0; j < EM_DL; ++j) G[hh + 1][j] += wc * g[j];
        }
#pragma unroll
        for (int j = 0; j < EM_DL; ++j) G[0][j] += g[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // dZ_i = ds_i q_vec (1 - t_i^2) with t recomputed; A_h += c_ih dZ_i, A_0 += dZ_i, d(q_vec) += ds_i t_i   (lanes = columns)
        for (int i = 0; i < a.S; ++i) {
            const float ds = dsb[i];
#pragma unroll
            for (int j = 0; j < EM_QL; ++j) {
                if (lane + 64 * j < a.q) {
                    float z = vreg[0][j];
#pragma unroll
                    for (int hh = 0; hh < EM_HMAX; ++hh)
                        if (hh < a.h) z += cs[i][hh] * vreg[hh + 1][j];
                    const float t = tanhf(z);
                    const float dz = ds * qv[j] * (1.0f - t * t);
                    dqv[j] += ds * t;
                    A[0][j] += dz;
#pragma unroll
                    for (int hh = 0; hh < EM_HMAX; ++hh)
                        if (hh < a.h) A[hh + 1][j] += cs[i][hh] * dz;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // this wave's share: [A_0 .. A_h | d(q_vec) | G_0 .. G_h]
    float* out = a.partial + ((long)blockIdx.x * EM_WPB + wave) * ((long)(a.h + 1) * a.q + a.q + (long)(a.h + 1) * a.d);
#pragma unroll
    for (int hh = 0; hh <= EM_HMAX; ++hh) {
        if (hh <= a.h) {
#pragma unroll
            for (int j = 0; j < EM_QL; ++j) { const int n = lane + 64 * j; if (n < a.q) out[(long)hh * a.q + n] = A[hh][j]; }
        }
    }
    float* o2 = out + (long)(a.h + 1) * a.q;
#pragma unroll
    for (int j = 0; j < EM_QL; ++j) { const int n = lane + 64 * j; if (n < a.q) o2[n] = dqv[j]; }
    float* o3 = o2 + a.q;
#pragma unroll
    for (int hh = 0; hh <= EM_HMAX; ++hh) {
        if (hh <= a.h) {
#pragma unroll
            for (int j = 0; j < EM_DL; ++j) { const int o = lane + 64 * j; if (o < a.d) o3[(long)hh * a.d + o] = G[hh][j]; }
        }
    }
}

// red[e] = sum over the waves' shares, ascending
__global__ __launch_bounds__(256) void empty_reduce_kernel(long n_elems, int n_waves, const float* partial, float* red) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elems) return;
    float acc = 0.f;
    for (int w = 0; w < n_waves; ++w) acc += partial[(long)w * n_elems + e];
    red[e] = acc;
}

// D_h = G_h + W_add^T A_h (kept in `D` for the rank-1 updates); the bias gradients.  One workgroup.
__global__ __launch_bounds__(1024) void empty_finish_kernel(int d, int h, int dk, int q, const float* red, const float* w_add, const float* w_o,
                                                            float* D, float* db_o, float* db_add, float* dq_vec, float* db_v) {
    const float* A = red;
    const float* dqv = red + (long)(h + 1) * q;
    const float* G = dqv + q;
    for (int i = threadIdx.x; i < (h + 1) * d; i += 1024) {
        const int hh = i / d, o = i - hh * d;
        float acc = G[i];
        for (int n = 0; n < q; ++n) acc += w_add[(long)n * d + o] * A[(long)hh * q + n];
        D[i] = acc;
        if (hh == 0) db_o[o] += acc;
    }
    for (int n = threadIdx.x; n < q; n += 1024) { db_add[n] += A[n]; dq_vec[n] += dqv[n]; }
    __threadfence_block();
    __syncthreads();
    for (int c = threadIdx.x; c < h * dk; c += 1024) {
        const int hh = c / dk;
        float acc = 0.f;
        for (int o = 0; o < d; ++o) acc += w_o[(long)o * d + c] * D[(long)(hh + 1) * d + o];
        db_v[c] += acc;
    }
}

// d(W_add)[n][o] += sum_h A_h[n] u_h[o] + A_0[n] b_O[o];   d(W_O)[o][c] += D_{head(c)}[o] b_v[c]
__global__ __launch_bounds__(256) void empty_update_kernel(int d, int h, int dk, int q, const float* red, const float* D, const float* consts,
                                                           const float* b_o, const float* b_v, float* dw_add, float* dw_o) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n_add = (long)q * d;
    if (i < n_add) {
        const int n = (int)(i / d), o = (int)(i - (long)n * d);
        float acc = red[n] * b_o[o];
        for (int hh = 0; hh < h; ++hh) acc += red[(long)(hh + 1) * q + n] * consts[(long)hh * d + o];
        dw_add[i] += acc;
    } else if (i < n_add + (long)d * d) {
        const long k = i - n_add;
        const int o = (int)(k / d), c = (int)(k - (long)o * d);
        if (c < h * dk) dw_o[k] += D[(long)(c / dk + 1) * d + o] * b_v[c];
    }
}

struct EmptyWs { size_t consts, partial, red, D, total; int n_wg; long share; };
static size_t em_up256(size_t x) { return (x + 255) / 256 * 256; }
static EmptyWs empty_layout(const nrms_encoder_desc* d) {
    EmptyWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = em_up256(off + bytes); return o; };
    const int h = d->n_heads;
    w.share = (long)(h + 1) * d->q_dim + d->q_dim + (long)(h + 1) * d->d_model;
    const int want = (d->n_seq + EM_WPB - 1) / EM_WPB;
    w.n_wg = want < 1 ? 1 : (want > 256 ? 256 : want);
    w.consts = take(((size_t)h * d->d_model + (size_t)(h + 1) * d->q_dim) * 4);
    w.partial = take((size_t)w.n_wg * EM_WPB * w.share * 4);
    w.red = take((size_t)w.share * 4);
    w.D = take((size_t)(h + 1) * d->d_model * 4);
    w.total = off;
    return w;
}

static int empty_validate(const nrms_encoder_desc* d, const char* who) {
    NRMS_REQUIRE(d != nullptr, "%s: null desc", who);
    NRMS_REQUIRE(d->n_seq >= 0 && d->seq_len >= 1 && d->seq_len <= 64, "%s: n_seq=%d seq_len=%d (1 .. 64)", who, d->n_seq, d->seq_len);
    NRMS_REQUIRE(d->use_output_proj != 0 && d->mask_mode == 0 && d->p_drop_embed == 0.f && d->p_drop_ctx == 0.f,
                 "%s: the closed form covers the output-projection topology without masks, embedding or context dropout", who);
    NRMS_REQUIRE(d->n_heads >= 1 && d->n_heads <= EM_HMAX && d->d_model > 0 && d->d_model % d->n_heads == 0 && d->d_model <= 64 * EM_DL &&
                 d->q_dim > 0 && d->q_dim <= 64 * EM_QL, "%s: d=%d h=%d q=%d (h <= %d, d <= %d, q <= %d)", who, d->d_model, d->n_heads,
                 d->q_dim, EM_HMAX, 64 * EM_DL, 64 * EM_QL);
    NRMS_REQUIRE(d->p_drop_attn >= 0.f && d->p_drop_attn < 1.f, "%s: p_drop_attn=%g", who, (double)d->p_drop_attn);
    return NRMS_OK;
}

static int empty_args(EmptyArgs* a, const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int32_t* seq_index, char* base,
                      const EmptyWs& L, hipStream_t s) {
    a->n_seq = desc->n_seq; a->S = desc->seq_len; a->d = desc->d_model; a->h = desc->n_heads; a->dk = desc->d_model / desc->n_heads; a->q = desc->q_dim;
    a->seq_index = seq_index;
    a->pd = make_dropout(desc->seed, desc->p_drop_attn);
    a->consts = (const float*)(base + L.consts);
    a->b_o = w->b_o; a->q_vec = w->q_vec;
    hipLaunchKernelGGL(empty_consts_kernel, dim3(1), dim3(1024), 0, s, a->d, a->h, a->dk, a->q, w->b_qkv + 2 * (long)a->d, w->w_o, w->b_o, w->w_add,
                       w->b_add, (float*)(base + L.consts));
    return check_launch("encoder_empty(consts)");
}

}  // namespace nrms

using namespace nrms;

extern "C" size_t nrms_encoder_empty_workspace_bytes(const nrms_encoder_desc* desc) {
    if (empty_validate(desc, "encoder_empty_workspace_bytes")) return 0;
    return empty_layout(desc).total;
}

extern "C" int nrms_encoder_empty_fwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int32_t* seq_index, float* out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    int rc = empty_validate(desc, "encoder_empty_fwd");
    if (rc) return rc;
    if (desc->n_seq == 0) return NRMS_OK;
    NRMS_REQUIRE(w && w->b_qkv && w->w_o && w->b_o && w->w_add && w->b_add && w->q_vec && out, "encoder_empty_fwd: null argument");
    const EmptyWs L = empty_layout(desc);
    if (workspace == nullptr || workspace_bytes < L.total) { set_error("encoder_empty_fwd: workspace %zu < required %zu bytes", workspace_bytes, L.total); return NRMS_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("empty_seq_fwd", s);
    EmptyArgs a{};
    rc = empty_args(&a, desc, w, seq_index, (char*)workspace, L, s);
    if (rc) return rc;
    a.out = out;
    hipLaunchKernelGGL(empty_fwd_kernel, dim3(L.n_wg), dim3(64 * EM_WPB), 0, s, a);
    return check_launch("encoder_empty_fwd");
}

extern "C" int nrms_encoder_empty_bwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int32_t* seq_index,
                                      const float* dout, const nrms_encoder_grads* g, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = empty_validate(desc, "encoder_empty_bwd");
    if (rc) return rc;
    if (desc->n_seq == 0) return NRMS_OK;
    NRMS_REQUIRE(w && w->b_qkv && w->w_o && w->b_o && w->w_add && w->b_add && w->q_vec && dout, "encoder_empty_bwd: null argument");
    NRMS_REQUIRE(g && g->b_qkv && g->w_o && g->b_o && g->w_add && g->b_add && g->q_vec, "encoder_empty_bwd: null gradient buffer");
    const EmptyWs L = empty_layout(desc);
    if (workspace == nullptr || workspace_bytes < L.total) { set_error("encoder_empty_bwd: workspace %zu < required %zu bytes", workspace_bytes, L.total); return NRMS_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)workspace;
    TimingScope ts("empty_seq_bwd", s);
    EmptyArgs a{};
    rc = empty_args(&a, desc, w, seq_index, base, L, s);
    if (rc) return rc;
    a.dout = dout;
    a.partial = (float*)(base + L.partial);
    const int d = a.d, h = a.h, dk = a.dk, q = a.q;
    float* red = (float*)(base + L.red);
    float* D = (float*)(base + L.D);
    hipLaunchKernelGGL(empty_bwd_kernel, dim3(L.n_wg), dim3(64 * EM_WPB), 0, s, a);
    hipLaunchKernelGGL(empty_reduce_kernel, dim3((unsigned)((L.share + 255) / 256)), dim3(256), 0, s, L.share, L.n_wg * EM_WPB, (const float*)a.partial, red);
    hipLaunchKernelGGL(empty_finish_kernel, dim3(1), dim3(1024), 0, s, d, h, dk, q, (const float*)red, w->w_add, w->w_o, D, g->b_o, g->b_add, g->q_vec,
                       g->b_qkv + 2 * (long)d);
    const long n_upd = (long)q * d + (long)d * d;
    hipLaunchKernelGGL(empty_update_kernel, dim3((unsigned)((n_upd + 255) / 256)), dim3(256), 0, s, d, h, dk, q, (const float*)red, (const float*)D,
                       a.consts, w->b_o, w->b_qkv + 2 * (long)d, g->w_add, g->w_o);
    return check_launch("encoder_empty_bwd");
}
