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
// One wave per sequence, fp32 FMA arithmetic; partial sums per workgroup, added in a fixed order.
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
    float* partial;                 // backward [n_workgroups][(h + 1) q + q]
    float* saved;                   // [n_seq][64][EM_HMAX + 1]: c_i0 .. c_i7, w_i of every row (forward writes, backward reads; null: inference)
    float* saved_wc;                // [n_seq][EM_HMAX]: sum_i w_i c_ih
};

// consts: u_h = W_O[:, head h] b_v^(h);  v_h = W_add u_h (h = 1 .. H), v_0 = W_add b_O + b_add
__global__ __launch_bounds__(256) void empty_consts_u_kernel(int d, int h, int dk, const float* b_v, const float* w_o, float* u) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * d) return;
    const int hh = i / d, o = i - hh * d;
    const float* row = w_o + (long)o * d + hh * dk;
    float a0 = 0.f, a1 = 0.f;
    int j = 0;
    for (; j + 2 <= dk; j += 2) { a0 += row[j] * b_v[hh * dk + j]; a1 += row[j + 1] * b_v[hh * dk + j + 1]; }
    if (j < dk) a0 += row[j] * b_v[hh * dk + j];
    u[i] = a0 + a1;
}
__global__ __launch_bounds__(1024) void empty_consts_v_kernel(int d, int h, int q, const float* u, const float* b_o, const float* w_add, const float* b_add,
                                                              float* v) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 6);                   // a wave per output: coalesced rows of W_add
    if (i >= (h + 1) * q) return;
    const int hh = i / q, n = i - hh * q;
    const float* x = hh == 0 ? b_o : u + (long)(hh - 1) * d;
    float acc = 0.f;
    for (int o = lane; o < d; o += 64) acc += w_add[(long)n * d + o] * x[o];
    acc = wave_sum(acc);
    if (lane == 0) v[i] = acc + (hh == 0 ? b_add[n] : 0.f);
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

// tanh through one exponential: absolute error ~1e-7 (the rows it feeds are softmax logits and (1 - t^2) factors)
__device__ __forceinline__ float em_tanh(float z) { return 1.0f - 2.0f / (1.0f + __expf(2.0f * z)); }

// shared by both directions: c (LDS [64][EM_HMAX], also returned for this lane's row), s_i -> w_i of this lane's row
template <int HT>
__device__ __forceinline__ void empty_rows(const EmptyArgs& a, long seq, int lane, float (*cs)[EM_HMAX], const float (&vreg)[EM_HMAX + 1][EM_QL],
                                           const float (&qv)[EM_QL], float (&c)[EM_HMAX], float& w_i) {
    const long sidx = a.seq_index != nullptr ? (long)a.seq_index[seq] : seq;
    // the S * h (row, head) factors, spread over the 64 lanes (a row per lane would leave 44 lanes idle on a 20-word title)
    for (int item = lane; item < 64 * EM_HMAX; item += 64) {
        const int i = item / EM_HMAX, hh = item - i * EM_HMAX;
        cs[i][hh] = (hh < a.h && i < a.S) ? empty_keep_factor(a.pd, sidx * a.h + hh, a.S, i) : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int hh = 0; hh < HT; ++hh) c[hh] = cs[lane][hh];
    float s_mine = -3.0e38f;
    for (int i = 0; i < a.S; ++i) {
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < EM_QL; ++j) {
            if (lane + 64 * j < a.q) {
                float z = vreg[0][j];
#pragma unroll
                for (int hh = 0; hh < HT; ++hh)
                    if (hh < a.h) z += cs[i][hh] * vreg[hh + 1][j];
                part += em_tanh(z) * qv[j];
            }
        }
        part = wave_sum(part);
        if (lane == i) s_mine = part;
    }
    const float m = wave_max(s_mine);
    const float e = lane < a.S ? __expf(s_mine - m) : 0.f;
    w_i = e / wave_sum(e);
}

template <int HT>
__device__ __forceinline__ void empty_load_consts(const EmptyArgs& a, int lane, float (&vreg)[EM_HMAX + 1][EM_QL], float (&qv)[EM_QL]) {
    const float* v = a.consts + (long)a.h * a.d;
#pragma unroll
    for (int j = 0; j < EM_QL; ++j) {
        const int n = lane + 64 * j;
        qv[j] = n < a.q ? a.q_vec[n] : 0.f;
#pragma unroll
        for (int hh = 0; hh <= HT; ++hh) vreg[hh][j] = (n < a.q && hh <= a.h) ? v[(long)hh * a.q + n] : 0.f;
    }
}

template <int HT>
__global__ __launch_bounds__(64 * EM_WPB) void empty_fwd_kernel(EmptyArgs a) {
    __shared__ float cs_all[EM_WPB][64][EM_HMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*cs)[EM_HMAX] = cs_all[wave];
    float vreg[EM_HMAX + 1][EM_QL], qv[EM_QL];
    empty_load_consts<HT>(a, lane, vreg, qv);
    for (long seq = (long)blockIdx.x * EM_WPB + wave; seq < a.n_seq; seq += (long)gridDim.x * EM_WPB) {
        float c[EM_HMAX], w_i;
        empty_rows<HT>(a, seq, lane, cs, vreg, qv, c, w_i);
        float wc[EM_HMAX];
#pragma unroll
        for (int hh = 0; hh < HT; ++hh) wc[hh] = hh < a.h ? wave_sum(w_i * c[hh]) : 0.f;
#pragma unroll
        for (int j = 0; j < EM_DL; ++j) {
            const int o = lane + 64 * j;
            if (o < a.d) {
                float y = a.b_o[o];
#pragma unroll
                for (int hh = 0; hh < HT; ++hh)
                    if (hh < a.h) y += wc[hh] * a.consts[(long)hh * a.d + o];
                a.out[seq * a.d + o] = y;
            }
        }
        if (a.saved != nullptr) {
            // for the backward: [c_i0 .. c_i7 | w_i] per row, [sum_i w_i c_ih] per sequence (Philox and the softmax are not redone)
            float* sv = a.saved + (seq * 64 + lane) * (EM_HMAX + 1);
#pragma unroll
            for (int hh = 0; hh < EM_HMAX; ++hh) sv[hh] = hh < HT ? c[hh] : 0.f;
            sv[EM_HMAX] = w_i;
            if (lane < EM_HMAX) {
                float v = 0.f;
#pragma unroll
                for (int hh = 0; hh < HT; ++hh) v = lane == hh ? wc[hh] : v;
                a.saved_wc[seq * EM_HMAX + lane] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();                                 // cs is rewritten by the next sequence
    }
}

// A_h += c_ih dZ_i, A_0 += dZ_i, d(q_vec) += ds_i t_i over this workgroup's sequences (t recomputed from the saved factors)
template <int HT>
__global__ __launch_bounds__(64 * EM_WPB) void empty_bwd_kernel(EmptyArgs a) {
    __shared__ float cs_all[EM_WPB][64][EM_HMAX];
    __shared__ float ds_all[EM_WPB][64];
    __shared__ float acc_s[(EM_HMAX + 1) * 64 * EM_QL + 64 * EM_QL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*cs)[EM_HMAX] = cs_all[wave];
    float* dsb = ds_all[wave];
    float vreg[EM_HMAX + 1][EM_QL], qv[EM_QL];
    empty_load_consts<HT>(a, lane, vreg, qv);
    float A[EM_HMAX + 1][EM_QL], dqv[EM_QL];
#pragma unroll
    for (int hh = 0; hh <= HT; ++hh)
#pragma unroll
        for (int j = 0; j < EM_QL; ++j) A[hh][j] = 0.f;
#pragma unroll
    for (int j = 0; j < EM_QL; ++j) dqv[j] = 0.f;
    for (long seq = (long)blockIdx.x * EM_WPB + wave; seq < a.n_seq; seq += (long)gridDim.x * EM_WPB) {
        const float* sv = a.saved + (seq * 64 + lane) * (EM_HMAX + 1);
        float c[EM_HMAX];
#pragma unroll
        for (int hh = 0; hh < HT; ++hh) { c[hh] = sv[hh]; cs[lane][hh] = c[hh]; }
        const float w_i = sv[EM_HMAX];
        // <g, u_h>, <g, b_O> with g = d(out) of this sequence
        float gu[EM_HMAX + 1];
#pragma unroll
        for (int hh = 0; hh <= HT; ++hh) gu[hh] = 0.f;
#pragma unroll
        for (int j = 0; j < EM_DL; ++j) {
            const int o = lane + 64 * j;
            if (o < a.d) {
                const float g = a.dout[seq * a.d + o];
                gu[0] += g * a.b_o[o];
#pragma unroll
                for (int hh = 0; hh < HT; ++hh)
                    if (hh < a.h) gu[hh + 1] += g * a.consts[(long)hh * a.d + o];
            }
        }
#pragma unroll
        for (int hh = 0; hh <= HT; ++hh) gu[hh] = hh <= a.h ? wave_sum(gu[hh]) : 0.f;
        // softmax backward over the rows (lane = row)
        float dw = gu[0];
#pragma unroll
        for (int hh = 0; hh < HT; ++hh) dw += c[hh] * gu[hh + 1];
        const float sumwd = wave_sum(w_i * dw);
        dsb[lane] = w_i * (dw - sumwd);                                  // 0 beyond the sequence (w_i = 0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // dZ_i = ds_i q_vec (1 - t_i^2); lanes = columns
        for (int i = 0; i < a.S; ++i) {
            const float ds = dsb[i];
#pragma unroll
            for (int j = 0; j < EM_QL; ++j) {
                if (lane + 64 * j < a.q) {
                    float z = vreg[0][j];
#pragma unroll
                    for (int hh = 0; hh < HT; ++hh)
                        if (hh < a.h) z += cs[i][hh] * vreg[hh + 1][j];
                    const float t = em_tanh(z);
                    const float dz = ds * qv[j] * (1.0f - t * t);
                    dqv[j] += ds * t;
                    A[0][j] += dz;
#pragma unroll
                    for (int hh = 0; hh < HT; ++hh)
                        if (hh < a.h) A[hh + 1][j] += cs[i][hh] * dz;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the workgroup's share [A_0 .. A_h | d(q_vec)]: its four waves add theirs in wave order (LDS), one row per workgroup
    const long share = (long)(a.h + 1) * a.q + a.q;
    for (int wv = 0; wv < EM_WPB; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int hh = 0; hh <= HT; ++hh) {
                if (hh <= a.h) {
#pragma unroll
                    for (int j = 0; j < EM_QL; ++j) {
                        const int n = lane + 64 * j;
                        if (n < a.q) { float* o = acc_s + (long)hh * a.q + n; *o = wv == 0 ? A[hh][j] : *o + A[hh][j]; }
                    }
                }
            }
            float* o2 = acc_s + (long)(a.h + 1) * a.q;
#pragma unroll
            for (int j = 0; j < EM_QL; ++j) { const int n = lane + 64 * j; if (n < a.q) o2[n] = wv == 0 ? dqv[j] : o2[n] + dqv[j]; }
        }
        __syncthreads();
    }
    float* out = a.partial + (long)blockIdx.x * share;
    for (long e = threadIdx.x; e < share; e += 64 * EM_WPB) out[e] = acc_s[e];
}

// G_0 = sum_seq g, G_h = sum_seq (sum_i w_i c_ih) g: workgroup b adds the sequences of its chunk in order, thread = column
__global__ __launch_bounds__(512) void empty_g_kernel(int n_seq, int chunk, int d, int h, const float* wc, const float* dout, float* partial) {
    const int o = threadIdx.x;
    const int s0 = blockIdx.x * chunk, s1 = min(s0 + chunk, n_seq);
    float acc[EM_HMAX + 1];
#pragma unroll
    for (int hh = 0; hh <= EM_HMAX; ++hh) acc[hh] = 0.f;
    if (o < d) {
        for (int s = s0; s < s1; ++s) {
            const float g = dout[(long)s * d + o];
            acc[0] += g;
#pragma unroll
            for (int hh = 0; hh < EM_HMAX; ++hh) acc[hh + 1] += wc[(long)s * EM_HMAX + hh] * g;
        }
#pragma unroll
        for (int hh = 0; hh <= EM_HMAX; ++hh)
            if (hh <= h) partial[((long)blockIdx.x * (h + 1) + hh) * d + o] = acc[hh];
    }
}

// red[e] = sum over the workgroups' shares in a fixed association: four row quarters (four partial sums each, rows ascending), added
// in quarter order.  Thread = (element, quarter): 64 elements per workgroup.
__global__ __launch_bounds__(256) void empty_reduce_kernel(long n_elems, int n_rows, const float* partial, float* red) {
    __shared__ float part[4][64];
    const int el = threadIdx.x & 63, qt = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + el;
    const int per = (n_rows + 3) / 4;
    const int r0 = qt * per, r1 = min(r0 + per, n_rows);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < n_elems) {
        int r = r0;
        for (; r + 4 <= r1; r += 4) {
            a0 += partial[(long)r * n_elems + e];
            a1 += partial[(long)(r + 1) * n_elems + e];
            a2 += partial[(long)(r + 2) * n_elems + e];
            a3 += partial[(long)(r + 3) * n_elems + e];
        }
        for (; r < r1; ++r) a0 += partial[(long)r * n_elems + e];
    }
    part[qt][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (qt == 0 && e < n_elems) red[e] = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
}

// D_h = G_h + W_add^T A_h (kept in `D` for the rank-1 updates), d(b_O) += D_0: workgroup = h, thread = column
__global__ __launch_bounds__(512) void empty_d_kernel(int d, int q, const float* red, long g_off, const float* w_add, float* D, float* db_o) {
    const int hh = blockIdx.x, o = threadIdx.x;
    if (o >= d) return;
    const float* A = red + (long)hh * q;
    float a0 = red[g_off + (long)hh * d + o], a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int n = 0;
    for (; n + 4 <= q; n += 4) {
        a0 += w_add[(long)n * d + o] * A[n];
        a1 += w_add[(long)(n + 1) * d + o] * A[n + 1];
        a2 += w_add[(long)(n + 2) * d + o] * A[n + 2];
        a3 += w_add[(long)(n + 3) * d + o] * A[n + 3];
    }
    for (; n < q; ++n) a0 += w_add[(long)n * d + o] * A[n];
    const float acc = (a0 + a1) + (a2 + a3);
    D[(long)hh * d + o] = acc;
    if (hh == 0) db_o[o] += acc;
}
// d(b_add) += A_0, d(q_vec) += sum ds t, d(b_v^(h)) += W_O[:, head h]^T D_h.  One workgroup.
__global__ __launch_bounds__(512) void empty_finish_kernel(int d, int h, int dk, int q, const float* red, const float* w_o, const float* D, float* db_add,
                                                           float* dq_vec, float* db_v) {
    const float* dqv = red + (long)(h + 1) * q;
    for (int n = threadIdx.x; n < q; n += 512) { db_add[n] += red[n]; dq_vec[n] += dqv[n]; }
    for (int c = threadIdx.x; c < h * dk; c += 512) {
        const float* Dh = D + (long)(c / dk + 1) * d;
        float a0 = 0.f, a1 = 0.f;
        int o = 0;
        for (; o + 2 <= d; o += 2) { a0 += w_o[(long)o * d + c] * Dh[o]; a1 += w_o[(long)(o + 1) * d + c] * Dh[o + 1]; }
        if (o < d) a0 += w_o[(long)o * d + c] * Dh[o];
        db_v[c] += a0 + a1;
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

struct EmptyWs { size_t consts, partial, partial_g, red, D, total; int n_wg, n_wg_fwd, n_chunks, chunk; long share, share_a; };
static size_t em_up256(size_t x) { return (x + 255) / 256 * 256; }
static EmptyWs empty_layout(const nrms_encoder_desc* d) {
    EmptyWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = em_up256(off + bytes); return o; };
    const int h = d->n_heads;
    w.share = (long)(h + 1) * d->q_dim + d->q_dim + (long)(h + 1) * d->d_model;
    const int want = (d->n_seq + EM_WPB - 1) / EM_WPB;
    w.n_wg = want < 1 ? 1 : (want > 1024 ? 1024 : want);                  // backward: one partial row per workgroup
    w.n_wg_fwd = want < 1 ? 1 : (want > 4096 ? 4096 : want);
    w.consts = take(((size_t)h * d->d_model + (size_t)(h + 1) * d->q_dim) * 4);
    w.share_a = (long)(h + 1) * d->q_dim + d->q_dim;
    w.n_chunks = d->n_seq < 256 ? (d->n_seq > 0 ? d->n_seq : 1) : 256;
    w.chunk = (d->n_seq + w.n_chunks - 1) / w.n_chunks;
    if (w.chunk < 1) w.chunk = 1;
    w.partial = take((size_t)w.n_wg * w.share_a * 4);
    w.partial_g = take((size_t)w.n_chunks * (h + 1) * d->d_model * 4);
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
    float* u = (float*)(base + L.consts);
    hipLaunchKernelGGL(empty_consts_u_kernel, dim3(cdiv((long)a->h * a->d, 256)), dim3(256), 0, s, a->d, a->h, a->dk, w->b_qkv + 2 * (long)a->d, w->w_o, u);
    hipLaunchKernelGGL(empty_consts_v_kernel, dim3(cdiv((long)(a->h + 1) * a->q, 16)), dim3(1024), 0, s, a->d, a->h, a->q, (const float*)u, w->b_o, w->w_add,
                       w->b_add, u + (long)a->h * a->d);
    return check_launch("encoder_empty(consts)");
}

}  // namespace nrms

using namespace nrms;

extern "C" size_t nrms_encoder_empty_workspace_bytes(const nrms_encoder_desc* desc) {
    if (empty_validate(desc, "encoder_empty_workspace_bytes")) return 0;
    return empty_layout(desc).total;
}

extern "C" size_t nrms_encoder_empty_saved_bytes(const nrms_encoder_desc* desc) {
    if (empty_validate(desc, "encoder_empty_saved_bytes")) return 0;
    return ((size_t)desc->n_seq * 64 * (EM_HMAX + 1) + (size_t)desc->n_seq * EM_HMAX) * sizeof(float);
}

extern "C" int nrms_encoder_empty_fwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int32_t* seq_index, float* out,
                                      void* saved, void* workspace, size_t workspace_bytes, void* stream) {
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
    a.saved = (float*)saved;
    a.saved_wc = saved != nullptr ? (float*)saved + (size_t)desc->n_seq * 64 * (EM_HMAX + 1) : nullptr;
    switch ((a.h + 1) / 2) {
        case 1: hipLaunchKernelGGL(empty_fwd_kernel<2>, dim3(L.n_wg_fwd), dim3(64 * EM_WPB), 0, s, a); break;
        case 2: hipLaunchKernelGGL(empty_fwd_kernel<4>, dim3(L.n_wg_fwd), dim3(64 * EM_WPB), 0, s, a); break;
        case 3: hipLaunchKernelGGL(empty_fwd_kernel<6>, dim3(L.n_wg_fwd), dim3(64 * EM_WPB), 0, s, a); break;
        default: hipLaunchKernelGGL(empty_fwd_kernel<8>, dim3(L.n_wg_fwd), dim3(64 * EM_WPB), 0, s, a); break;
    }
    return check_launch("encoder_empty_fwd");
}

extern "C" int nrms_encoder_empty_bwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int32_t* seq_index,
                                      const float* dout, const void* saved, const nrms_encoder_grads* g, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    int rc = empty_validate(desc, "encoder_empty_bwd");
    if (rc) return rc;
    if (desc->n_seq == 0) return NRMS_OK;
    NRMS_REQUIRE(w && w->b_qkv && w->w_o && w->b_o && w->w_add && w->b_add && w->q_vec && dout, "encoder_empty_bwd: null argument");
    NRMS_REQUIRE(saved != nullptr, "encoder_empty_bwd: `saved` (what nrms_encoder_empty_fwd wrote for these sequences) is required");
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
    a.saved = (float*)saved;
    a.saved_wc = (float*)saved + (size_t)desc->n_seq * 64 * (EM_HMAX + 1);
    float* part_g = (float*)(base + L.partial_g);
    switch ((h + 1) / 2) {
        case 1: hipLaunchKernelGGL(empty_bwd_kernel<2>, dim3(L.n_wg), dim3(64 * EM_WPB), 0, s, a); break;
        case 2: hipLaunchKernelGGL(empty_bwd_kernel<4>, dim3(L.n_wg), dim3(64 * EM_WPB), 0, s, a); break;
        case 3: hipLaunchKernelGGL(empty_bwd_kernel<6>, dim3(L.n_wg), dim3(64 * EM_WPB), 0, s, a); break;
        default: hipLaunchKernelGGL(empty_bwd_kernel<8>, dim3(L.n_wg), dim3(64 * EM_WPB), 0, s, a); break;
    }
    hipLaunchKernelGGL(empty_g_kernel, dim3(L.n_chunks), dim3(512), 0, s, desc->n_seq, L.chunk, d, h, (const float*)a.saved_wc, dout, part_g);
    hipLaunchKernelGGL(empty_reduce_kernel, dim3((unsigned)((L.share_a + 63) / 64)), dim3(256), 0, s, L.share_a, L.n_wg, (const float*)a.partial, red);
    const long share_g = (long)(h + 1) * d;
    hipLaunchKernelGGL(empty_reduce_kernel, dim3((unsigned)((share_g + 63) / 64)), dim3(256), 0, s, share_g, L.n_chunks, (const float*)part_g, red + L.share_a);
    hipLaunchKernelGGL(empty_d_kernel, dim3(h + 1), dim3(512), 0, s, d, q, (const float*)red, L.share_a, w->w_add, D, g->b_o);
    hipLaunchKernelGGL(empty_finish_kernel, dim3(1), dim3(512), 0, s, d, h, dk, q, (const float*)red, w->w_o, (const float*)D, g->b_add, g->q_vec,
                       g->b_qkv + 2 * (long)d);
    const long n_upd = (long)q * d + (long)d * d;
    hipLaunchKernelGGL(empty_update_kernel, dim3((unsigned)((n_upd + 255) / 256)), dim3(256), 0, s, d, h, dk, q, (const float*)red, (const float*)D,
                       a.consts, w->b_o, w->b_qkv + 2 * (long)d, g->w_add, g->w_o);
    return check_launch("encoder_empty_bwd");
}
