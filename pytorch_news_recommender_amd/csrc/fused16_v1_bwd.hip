// fp16 mode, backward of nrms_v1's news encoder (fused16_v1.hip): heads wider than 32 and the output projection W_O.
// Replaces autograd through model/nrms_v1.py:13-23,50-76,87-105,109-162 (`loss.backward()`, train_eval.py:126).
//
//   dout --(dout16: ten output blocks of d / 10 columns = the layout of ctx16)--> fused_bwd16_pool_kernel (fused16_bwd.hip,
//       unchanged: pooling backward, dZ16, d(o) = d(W_O output) after the dropout mask -> dctx16; all-padding titles in
//       closed form: their sum E of d(o) rows lands in the "V" slots of the column sums)
//   fused_bwd16v1_attn_kernel, one wave per long title / pair of short titles:
//       phase A  d(attn)^T = W_O^T d(o)^T as 2 h tiles of [32 features x tile rows] (feature blocks of the heads), parked in a
//                per-wave scratch (written and re-read by the same lanes; 48 MB for the whole grid: it lives in the caches);
//       phase B  per head, over the two feature blocks b: V_b^T, Q_b^T, K_b^T recomputed from x16 (six tiles), dP^T = sum_b V_b
//                d(attn_b)^T, S^T = sum_b K_b Q_b^T, P^T, dS^T, and per block dV_b, dK_b, dQ_b -> dqkv16 rows of
//                [head][Q0 K0 Q1 K1 V0 V1][32] columns (192 per head), bias column sums without atomics
//   gemm16_dx (18 slabs for six heads), gemm16_tn: d(W_qkv) = dQKV^T X, d(W_add) = dZ^T ctx, and
//       d(W_O) = d(o)^T attn over the 32-row blocks of the titles with a real token (a short title's compressed d(o) block against
//       its tile-row attn block is the same sum as token by token); the ones column of attn16 makes d(b_o) one more column of it
//   all-padding titles (every attn row = b_v): d(b_o) += E, d(W_O) += E (x) b_v, d(b_v) += W_O^T E.
#include <stdlib.h>

#include "fused16_bwd.h"
#include "fused16_v1.h"

namespace nrms {

constexpr int V1_RED = 36 * 32;               // per-workgroup column sums of the attention kernel: [head][Q0 K0 Q1 K1 V0 V1][32]
constexpr int V1_HMAX = 6;                    // 3 h + 1 <= 19 k-steps of attn16

struct Bwd16V1Args {
    int n_seq, S;
    V1Geom g;
    const _Float16* x16;
    const int* pos;
    const int* n_rows;
    const int64_t* ids;
    const int* order;           // [3][n_seq]: long | all-padding | short
    const int* order_cnt;
    const _Float16* btiles;     // [2 h W_O^T tiles | 6 h head tiles] (row-major [32][KP])
    const _Float16* dctx16;     // d(o): [n_seq][20][32][16], short titles compressed to rows 0 .. n
    _Float16* scratch;          // [gridDim.x * F16_WAVES][4 h][64][8]
    _Float16* dqkv16;           // [rows][192 h]
    float* red;                 // [gridDim.x][V1_RED]
};

__global__ __launch_bounds__(F16_THREADS, 2) void fused_bwd16v1_attn_kernel(Bwd16V1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + 3 * F16_SLOT);          // [V1_RED]
    float* stg = red + V1_RED;                                           // [2][F16_WAVES][6][32]: per-wave shares of one head's sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int S = a.S, H = a.g.h;
    const int ldq = 192 * H;
    constexpr int KP = F16_KP;
    constexpr float NEG = -3.0e38f;
    for (int i = tid; i < V1_RED; i += F16_THREADS) red[i] = 0.f;
    h8 idf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) idf[s][j] = (_Float16)(l32 == 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3) ? 1.0f : 0.0f);
    const int n_long = a.order_cnt[0], n_sh = a.order_cnt[2];
    const int g_long = (n_long + F16_WAVES - 1) / F16_WAVES, g_pair = (n_sh + 2 * F16_WAVES - 1) / (2 * F16_WAVES);
    TileRing ring;
    ring.smem = smem; ring.src = a.btiles; ring.n_tiles = 8 * H; ring.tid = tid; ring.l32 = l32; ring.hh = hh; ring.dbg = 0;
    _Float16* scr = a.scratch + ((long)blockIdx.x * F16_WAVES + wave) * (long)(4 * H * 512) + lane * 8;
    const h8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
#pragma unroll 1
    for (int grp = blockIdx.x; grp < g_pair + g_long; grp += gridDim.x) {
        const bool pair = grp < g_pair;                                 // uniform over the workgroup
        int seq2[2] = {0, 0};
        bool val2[2] = {false, false};
        int nl2[2] = {0, 0};
        if (pair) {
            const int p0 = 2 * (grp * F16_WAVES + wave);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                val2[i] = p0 + i < n_sh;
                seq2[i] = val2[i] ? a.order[2 * (long)a.n_seq + p0 + i] : 0;
                if (val2[i]) nl2[i] = __popcll(__ballot(lane < S && a.ids[(long)seq2[i] * S + lane] != 0));
            }
        } else {
            const int slot_id = (grp - g_pair) * F16_WAVES + wave;
            val2[0] = slot_id < n_long;
            seq2[0] = val2[0] ? a.order[slot_id] : 0;
        }
        const bool live = val2[0] || val2[1];
        const int myp = pair ? (l32 >> 4) : 0, myr = pair ? (l32 & 15) : l32;
        const int myn = myp ? nl2[1] : nl2[0];
        const bool myval = myp ? val2[1] : val2[0];
        const int myseq = myp ? seq2[1] : seq2[0];
        f32x16 kbias;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = crow32(r, hh);
            float bsv = j < S ? 0.f : NEG;
            if (pair) {
                const int jr = j & 15;
                bsv = (j >> 4) != myp ? NEG : (jr < myn ? 0.f : ((jr == myn && myn < S) ? __logf((float)max(S - myn, 1)) : NEG));
            }
            kbias[r] = bsv;
        }
        ring.load(0); ring.store(0);
        ring.load(1); ring.store(1);
        int n = 0;
        // ================= phase A: d(attn)^T = W_O^T d(o)^T, 2 h tiles -> scratch =================
        {
            h8 df[F16_CS];                                              // this tile row's d(o) (pair: row myr of title myp's compressed block)
            const _Float16* dsrc = a.dctx16 + frag_off((long)myseq, F16_CS, 0, myr, hh);
#pragma unroll
            for (int s = 0; s < F16_CS; ++s) df[s] = *reinterpret_cast<const h8*>(dsrc + 512 * s);
            if (!myval) {
#pragma unroll
                for (int s = 0; s < F16_CS; ++s) df[s] = z8;
            }
            __syncthreads();
#pragma unroll 1
            for (int t = 0; t < 2 * H; ++t) {
                ring.load(n + 2);
                f32x16 acc = zero16();
                if (live) {
                    tile_mma<true>(acc, ring, n, df);
                    *reinterpret_cast<h8*>(scr + (2 * t) * 512) = acc_frag(acc, 0);
                    *reinterpret_cast<h8*>(scr + (2 * t + 1) * 512) = acc_frag(acc, 1);
                }
                ring.store(n + 2);
                __syncthreads();
                ++n;
            }
        }
        // ================= phase B =================
        bool tok_ok;
        long drow;                                                      // x16 / dqkv16 row of this lane's token, -1: none
        h8 xf[F16_KS];
        {
            long xrow;
            if (pair) {
                tok_ok = myval && myr <= myn;
                drow = (myval && myr < myn) ? (long)a.pos[(long)myseq * S + myr] : -1;
                xrow = drow >= 0 ? drow : ((myval && myr == myn && myn < S) ? (long)*a.n_rows : -1);
            } else {
                tok_ok = myval && l32 < S;
                drow = tok_ok ? (long)a.pos[(long)myseq * S + l32] : -1;
                xrow = (tok_ok && drow < 0) ? (long)*a.n_rows : drow;
            }
            const _Float16* xr = a.x16 + (xrow < 0 ? 0 : xrow) * KP + 8 * hh;
#pragma unroll
            for (int s = 0; s < F16_KS; ++s) xf[s] = *reinterpret_cast<const h8*>(xr + 16 * s);
            if (xrow < 0) {
#pragma unroll
                for (int s = 0; s < F16_KS; ++s) xf[s] = z8;
            }
        }
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's scratch stores have landed
#pragma unroll 1
        for (int head = 0; head < H; ++head) {
            // tile order of a head: V0 V1 Q0 K0 Q1 K1 -- dP^T needs only V and d(attn), so it is formed first and the V fragments
            // never coexist with the Q | K fragments of both blocks (which must survive until dS^T exists)
            h8 dc[2][2];                                                // d(attn)^T of the head's two feature blocks, operand fragments
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) dc[b][c] = live ? *reinterpret_cast<const h8*>(scr + (4 * head + 2 * b + c) * 512) : z8;
            // dP^T[j][i] = sum_b sum_f V_b^T[f][j] d(attn_b)^T[f][i]
            f32x16 dst = zero16();
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                ring.load(n + 2);
                {
                    f32x16 t = zero16();
                    if (live) {
                        tile_mma<true>(t, ring, n, xf);
                        dst = mfma32h(acc_frag(t, 0), dc[b][0], dst);
                        dst = mfma32h(acc_frag(t, 1), dc[b][1], dst);
                    }
                }
                ring.store(n + 2);
                __syncthreads();
                ++n;
            }
            h8 qf[2][2], kf[2][2];
            f32x16 pt = zero16();
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                ring.load(n + 2);
                {
                    f32x16 t = zero16();
                    if (live) tile_mma<true>(t, ring, n, xf);
                    qf[b][0] = acc_frag(t, 0); qf[b][1] = acc_frag(t, 1);
                }
                ring.store(n + 2);
                __syncthreads();
                ++n;
                ring.load(n + 2);
                {
                    f32x16 t = zero16();
                    if (live) tile_mma<true>(t, ring, n, xf);
                    kf[b][0] = acc_frag(t, 0); kf[b][1] = acc_frag(t, 1);
                }
                ring.store(n + 2);
                if (b == 0) { __syncthreads(); ++n; }
                if (live) {
                    pt = mfma32h(kf[b][0], qf[b][0], pt);               // S^T[j][i] over this block's features
                    pt = mfma32h(kf[b][1], qf[b][1], pt);
                }
            }
            // d(attn)^T again, for dV (re-read from the scratch: cheaper than 16 registers held across the four tiles above)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) dc[b][c] = live ? *reinterpret_cast<const h8*>(scr + (4 * head + 2 * b + c) * 512) : z8;
            if (live) {
                float m = NEG;
#pragma unroll
                for (int r = 0; r < 16; ++r) { pt[r] += kbias[r]; m = fmaxf(m, pt[r]); }
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { pt[r] = __expf(pt[r] - m); sum += pt[r]; }
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int r = 0; r < 16; ++r) pt[r] *= inv;
            }
            float* srow = stg + (((head & 1) * F16_WAVES + wave) * 6) * 32 + l32;
            if (!live && hh == 0) {
#pragma unroll
                for (int t6 = 0; t6 < 6; ++t6) srow[32 * t6] = 0.f;
            }
            if (live) {
                float delta = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) delta += pt[r] * dst[r];
                delta += __shfl_xor(delta, 32, 64);
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[r] = pt[r] * (dst[r] - delta);
                const h8 pf[2] = {acc_frag(pt, 0), acc_frag(pt, 1)};
                const h8 sf[2] = {acc_frag(dst, 0), acc_frag(dst, 1)};
                _Float16* orow = a.dqkv16 + (drow < 0 ? 0 : drow) * (long)ldq + head * 192 + 8 * hh;
                // ---- dV_b^T[f][j] = sum_i d(attn_b)[i][f] P[i][j];  d(b_v) = sum_i d(attn)_i (rows of P sum to 1)
                {
                    const f32x16 p = transpose_frags(pf[0], pf[1], idf);                    // [i][j]
                    const h8 p0 = acc_frag(p, 0), p1 = acc_frag(p, 1);
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const f32x16 dctx = transpose_frags(dc[b][0], dc[b][1], idf);     // [i][f]
                        float sv = regsum(dctx);
                        sv += __shfl_xor(sv, 32, 64);
                        if (hh == 0) srow[32 * (4 + b)] = sv;
                        f32x16 dv = mfma32h(acc_frag(dctx, 0), p0, zero16());
                        dv = mfma32h(acc_frag(dctx, 1), p1, dv);
                        if (drow >= 0) {
                            *reinterpret_cast<h8*>(orow + 32 * (4 + b)) = acc_frag(dv, 0);
                            *reinterpret_cast<h8*>(orow + 32 * (4 + b) + 16) = acc_frag(dv, 1);
                        }
                    }
                }
                // ---- dK_b^T[f][j] = sum_i Q'_b[i][f] dS[i][j]
                {
                    const f32x16 dsn = transpose_frags(sf[0], sf[1], idf);                  // [i][j]
                    const h8 s0 = acc_frag(dsn, 0), s1 = acc_frag(dsn, 1);
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const f32x16 q = transpose_frags(qf[b][0], qf[b][1], idf);          // [i][f]
                        f32x16 dk = mfma32h(acc_frag(q, 0), s0, zero16());
                        dk = mfma32h(acc_frag(q, 1), s1, dk);
                        const h8 dk0 = acc_frag(dk, 0), dk1 = acc_frag(dk, 1);
                        if (drow >= 0) {
                            *reinterpret_cast<h8*>(orow + 32 * (2 * b + 1)) = dk0;
                            *reinterpret_cast<h8*>(orow + 32 * (2 * b + 1) + 16) = dk1;
                        }
                        float sk = regsum(transpose_frags(dk0, dk1, idf));
                        sk += __shfl_xor(sk, 32, 64);
                        if (hh == 0) srow[32 * (2 * b + 1)] = sk;
                    }
                }
                // ---- dQ'_b^T[f][i] = sum_j K_b[j][f] dS^T[j][i]
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const f32x16 k = transpose_frags(kf[b][0], kf[b][1], idf);              // [j][f]
                    f32x16 dq = mfma32h(acc_frag(k, 0), sf[0], zero16());
                    dq = mfma32h(acc_frag(k, 1), sf[1], dq);
                    const h8 dq0 = acc_frag(dq, 0), dq1 = acc_frag(dq, 1);
                    if (drow >= 0) {
                        *reinterpret_cast<h8*>(orow + 32 * (2 * b)) = dq0;
                        *reinterpret_cast<h8*>(orow + 32 * (2 * b) + 16) = dq1;
                    }
                    float sq = regsum(transpose_frags(dq0, dq1, idf));
                    sq += __shfl_xor(sq, 32, 64);
                    if (hh == 0) srow[32 * (2 * b)] = sq;
                }
            }
            __syncthreads();
            if (tid < 192) {                                              // fixed order: waves ascending
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < F16_WAVES; ++w) sum += stg[(((head & 1) * F16_WAVES + w) * 6) * 32 + tid];
                red[6 * head * 32 + tid] += sum;
            }
            ++n;
        }
        __syncthreads();                                          // the ring restarts: nobody may still read a slot
    }
    __syncthreads();
    float* out = a.red + (long)blockIdx.x * V1_RED;
    for (int i = tid; i < V1_RED; i += F16_THREADS) out[i] = red[i];
}

// ---------------------------------------------------------------------------------------------------------------
// column c (0 .. 192 h) of dqkv16 -> (which of Q | K | V, model feature) or -1 (padding of a feature block)
__host__ __device__ inline int v1_dq_col(const V1Geom& g, int m, int* which) {
    const int head = m / 192, rem = m - head * 192, t6 = rem >> 5, p = rem & 31;
    const int s = p >> 4, tt = p & 15, hh = tt >> 3, jj = tt & 7;
    const int f = 16 * s + 8 * (jj >> 2) + 4 * hh + (jj & 3);          // feature held at memory position p (acc_frag order)
    const int blk = t6 < 4 ? (t6 >> 1) : (t6 - 4);
    *which = t6 < 4 ? (t6 & 1) : 2;
    const int fh = 32 * blk + f;
    return (head < g.h && fh < g.dk) ? head * g.dk + fh : -1;
}

struct Prep16bV1Args {
    V1Geom g;
    int q;
    const float *w_qkv, *b_qkv, *w_o, *w_add, *q_vec;
    _Float16* btiles;     // [10 pool tiles | 2 h W_O^T tiles | 6 h head tiles][32][KP]
    _Float16* xtiles;     // [3 h slabs][4][10][2][32][8]
    _Float16* qv16;       // [QP]
};

__global__ __launch_bounds__(256) void prep16bv1_kernel(Prep16bV1Args a) {
    const V1Geom g = a.g;
    const float qscale = 1.0f / sqrtf((float)g.dk);
    constexpr int KP = F16_KP;
    const long n1 = (long)(10 + 8 * g.h) * 32 * KP, n2 = (long)3 * g.h * (DX_B_BYTES / 2), n3 = F16_QP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2 + n3; i += (long)gridDim.x * blockDim.x) {
        if (i < n1) {
            const int c = (int)(i % KP);
            const long r = i / KP;
            const int f = (int)(r & 31), tile = (int)(r >> 5);
            float v = 0.f;
            if (tile < 10) {                                    // Wadd_j^T over the ten output blocks: row f, column q
                if (f < g.dkv && c < a.q) v = a.w_add[(long)c * g.d + tile * g.dkv + f];
            } else if (tile < 10 + 2 * g.h) {                   // W_O^T: row = feature 32 b + f of head hd, column = position of d(o)
                const int t = tile - 10, hd = t >> 1, fh = 32 * (t & 1) + f;
                const int b16 = c >> 4, t16 = c & 15, hh = t16 >> 3, jj = t16 & 7;
                const int fpad = 16 * b16 + 8 * (jj >> 2) + 4 * hh + (jj & 3);
                const int j = fpad >> 5, fo = fpad & 31;
                if (fh < g.dk && fo < g.dkv && j < g.hv) v = a.w_o[(long)(j * g.dkv + fo) * g.d + hd * g.dk + fh];
            } else {
                // stream order of a head's tiles: V0 V1 Q0 K0 Q1 K1 (fused_bwd16v1_attn_kernel)
                const int t = tile - 10 - 2 * g.h, hd = t / 6, s6 = t - 6 * hd;
                const int which = s6 < 2 ? 2 : ((s6 - 2) & 1), blk = s6 < 2 ? s6 : ((s6 - 2) >> 1);
                const int fh = 32 * blk + f;
                if (fh < g.dk && c < g.d) v = a.w_qkv[((long)which * g.d + hd * g.dk + fh) * g.d + c] * (which == 0 ? qscale : 1.0f);
                if (fh < g.dk && c == g.d) v = a.b_qkv[which * g.d + hd * g.dk + fh] * (which == 0 ? qscale : 1.0f);   // ones column
            }
            a.btiles[i] = (_Float16)v;
        } else if (i < n1 + n2) {
            const long j = i - n1;
            // position j = ((((slab * 4 + s) * 10 + t) * 2 + hh) * 32 + l) * 8 + e  ->  W'[m][k], m = 64 slab + 16 s + 8 hh + e, k = 32 t + l
            const int e = (int)(j & 7), l = (int)((j >> 3) & 31), hh2 = (int)((j >> 8) & 1);
            long r = j >> 9;
            const int t = (int)(r % 10); r /= 10;
            const int s4 = (int)(r & 3), slab = (int)(r >> 2);
            const int m = 64 * slab + 16 * s4 + 8 * hh2 + e;     // dqkv16 column
            const int k = 32 * t + l;                            // input feature (output column of dX)
            int which;
            const int col = v1_dq_col(g, m, &which);
            float v = 0.f;
            if (col >= 0 && k < g.d) v = a.w_qkv[((long)which * g.d + col) * g.d + k] * (which == 0 ? qscale : 1.0f);
            a.xtiles[j] = (_Float16)v;
        } else {
            const long j = i - n1 - n2;
            a.qv16[j] = (_Float16)(j < a.q ? a.q_vec[j] : 0.f);
        }
    }
}

struct Maps16V1 { int *nmap_qkv, *kmap_x, *nmap_add, *kmap_ctx, *kmap_attn; float *nscale_qkv, *nscale_add, *nscale_o; };

__global__ void maps16v1_kernel(V1Geom g, int q, const float* sc, Maps16V1 m) {
    const float inv_scale = sc[1];
    const float qscale = 1.0f / sqrtf((float)g.dk);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 192 * g.h) {
        int which;
        const int col = v1_dq_col(g, i, &which);
        m.nmap_qkv[i] = col >= 0 ? which * g.d + col : -1;
        m.nscale_qkv[i] = inv_scale * (which == 0 ? qscale : 1.0f);
    }
    if (i < F16_KP) m.kmap_x[i] = i < g.d ? i : -1;
    if (i < F16_QP) { m.nmap_add[i] = i < q ? i : -1; m.nscale_add[i] = inv_scale; }
    if (i < F16_DP) {
        // ctx16 / dctx16 column -> output feature of W_O (ten blocks of d / 10)
        const int b16 = i >> 4, t = i & 15, hh = t >> 3, jj = t & 7;
        const int fpad = 16 * b16 + 8 * (jj >> 2) + 4 * hh + (jj & 3);
        const int j = fpad >> 5, f = fpad & 31;
        m.kmap_ctx[i] = (j < g.hv && f < g.dkv) ? j * g.dkv + f : -1;
        m.nscale_o[i] = inv_scale;
        // attn16 column -> input feature of W_O; the ones column -> the bias gradient (-2)
        const int ks = i >> 4, p = i & 15;
        int col = -1;
        if (i == V1_ONES_COL) col = -2;
        else if (ks < 3 * g.h) {
            const int hd = ks / 3, part = ks - 3 * hd, ph = p >> 3, pj = p & 7;
            const int fh = 16 * part + (((pj >> 2) << 3) | (ph << 2) | (pj & 3));
            if (fh < g.dk) col = hd * g.dk + fh;
        } else if (ks == g.kl && p < g.lo * g.h) {
            col = (p / g.lo) * g.dk + 48 + p % g.lo;
        }
        m.kmap_attn[i] = col;
    }
}

// column sums over the workgroups' rows (fixed order, as red16_kernel).  mode 0: the pool kernel's sums (pitch B16_RED): d(b_add),
// d(q_vec), and the "V" slots of the ten output blocks = E, the sum of d(o) over the rows of the all-padding titles -> ebuf[10][32];
// mode 1: the attention kernel's (pitch V1_RED) -> d(b_qkv)
__global__ __launch_bounds__(1024) void red16v1_kernel(int mode, const float* red, int n_wg, V1Geom g, int q, const float* sc,
                                                      float* db_qkv, float* db_add, float* dq_vec, float* ebuf) {
    const float inv_scale = sc[1];
    const int pitch = mode == 0 ? B16_RED : V1_RED, ncol = mode == 0 ? B16_RED : 192 * g.h;
    __shared__ float part[32][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float s = 0.f;
    if (i < ncol) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int w = rg;
        for (; w + 96 < n_wg; w += 128) {
            s0 += red[(long)w * pitch + i];
            s1 += red[(long)(w + 32) * pitch + i];
            s2 += red[(long)(w + 64) * pitch + i];
            s3 += red[(long)(w + 96) * pitch + i];
        }
        for (; w < n_wg; w += 32) s0 += red[(long)w * pitch + i];
        s = (s0 + s1) + (s2 + s3);
    }
    part[rg][c] = s;
    __syncthreads();
    if (rg != 0 || i >= ncol) return;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += part[r][c];
    s *= inv_scale;
    if (mode == 1) {
        // column i = [head][Q0 K0 Q1 K1 V0 V1][feature f of the block, natural order]
        const int head = i / 192, t6 = (i - head * 192) >> 5, f = i & 31;
        const int which = t6 < 4 ? (t6 & 1) : 2, fh = 32 * (t6 < 4 ? (t6 >> 1) : (t6 - 4)) + f;
        if (fh < g.dk) db_qkv[which * g.d + head * g.dk + fh] += s * (which == 0 ? 1.0f / sqrtf((float)g.dk) : 1.0f);
    } else if (i < B16_RED_QKV) {
        const int tile = i >> 5, f = i & 31, j = tile / 3, which = tile - 3 * j;
        if (which == 2) ebuf[j * 32 + f] = s;
    } else if (i < B16_RED_QKV + F16_QP) {
        const int qq = i - B16_RED_QKV;
        if (qq < q) db_add[qq] += s;
    } else {
        const int qq = i - B16_RED_QKV - F16_QP;
        if (qq < q) dq_vec[qq] += s;
    }
}

// all-padding titles, closed form: d(W_O)[o][c] += E[o] b_v[c], d(b_o)[o] += E[o]   (E without the loss scale)
__global__ __launch_bounds__(256) void closed16v1_wo_kernel(V1Geom g, const float* b_qkv, const float* ebuf, float* dw_o, float* db_o) {
    const int total = g.d * g.d;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int o = i / g.d, c = i - o * g.d;
        const float e = ebuf[(o / g.dkv) * 32 + o % g.dkv];
        if (e == 0.f) continue;
        dw_o[i] += e * b_qkv[2 * g.d + c];
        if (c == 0) db_o[o] += e;
    }
}
// ... and d(b_v)[c] += sum_o W_O[o][c] E[o]: 32 columns per workgroup, the o range dealt over 32 row groups (a single thread per
// column walked 300 dependent loads: 0.12 ms for a 90-kflop product)
__global__ __launch_bounds__(1024) void closed16v1_bv_kernel(V1Geom g, const float* w_o, const float* ebuf, float* db_qkv) {
    __shared__ float part[32][33];
    const int cl = threadIdx.x & 31, og = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < g.d)
        for (int o = og; o < g.d; o += 32) s += w_o[(long)o * g.d + c] * ebuf[(o / g.dkv) * 32 + o % g.dkv];
    part[og][cl] = s;
    __syncthreads();
    if (og != 0 || c >= g.d) return;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += part[r][cl];               // fixed order
    db_qkv[2 * g.d + c] += s;
}

// ---------------------------------------------------------------------------------------------------------------
static size_t up256(size_t x) { return (x + 255) / 256 * 256; }

struct Fused16V1BwdLayout {
    int n_wg, splits_qkv, splits_add, splits_o;
    size_t btiles, xtiles, qv16, dout16, dz16, dctx16, scratch, dqkv16, red_pool, red_attn, ebuf, maps, scale, p_qkv, p_add, p_o, total;
};

static Fused16V1BwdLayout v1_bwd_layout(long M, int n_seq, int h) {
    Fused16V1BwdLayout L;
    const long Mp = (long)n_seq * 32;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up256(off + bytes); return o; };
    L.n_wg = 512;
    L.btiles = take((size_t)(10 + 8 * h) * 32 * F16_KP * 2);
    L.xtiles = take((size_t)3 * h * DX_B_BYTES);
    L.qv16 = take((size_t)F16_QP * 2);
    L.dout16 = take((size_t)n_seq * F16_DP * 2);
    L.dz16 = take((size_t)Mp * F16_QP * 2);
    L.dctx16 = take((size_t)Mp * F16_DP * 2);
    L.scratch = take((size_t)L.n_wg * F16_WAVES * 4 * h * 1024);
    L.dqkv16 = take((size_t)(M + 32) * 192 * h * 2);
    L.red_pool = take((size_t)L.n_wg * B16_RED * 4);
    L.red_attn = take((size_t)L.n_wg * V1_RED * 4);
    L.ebuf = take((size_t)10 * 32 * 4);
    L.maps = take((size_t)(192 * V1_HMAX * 2 + F16_KP + F16_QP * 2 + F16_DP * 3) * 4);
    L.scale = take(256);
    // grids of at most 256 workgroups, multiples of 8 (see fused16_bwd_layout): d(W_qkv) has ceil(192 h / 320) x 2 output
    // blocks of 320 x 160, d(W_O) 1 x 2, d(W_add) one block of 224 x 320
    const int nb = cdiv(192 * h, 320) * 2;
    L.splits_qkv = 240 / nb;
    L.splits_add = 192;
    L.splits_o = 64;
    if (const char* e = getenv("NRMS_TN_SPLITS_QKV")) L.splits_qkv = atoi(e);      // tuning only
    if (const char* e = getenv("NRMS_TN_SPLITS_ADD")) L.splits_add = atoi(e);
    if (const char* e = getenv("NRMS_TN_SPLITS_O")) L.splits_o = atoi(e);
    L.p_qkv = take((size_t)L.splits_qkv * 192 * h * F16_KP * 4);
    L.p_add = take((size_t)L.splits_add * F16_QP * F16_DP * 4);
    L.p_o = take((size_t)L.splits_o * F16_DP * F16_DP * 4);
    L.total = off;
    return L;
}

size_t fused16v1_bwd_bytes(long M, int n_seq, int h) { return v1_bwd_layout(M, n_seq, h).total; }

int launch_fused_bwd16v1(const Fused16Bwd& f, hipStream_t stream) {
    if (f.n_seq <= 0) return NRMS_OK;
    if (f.order == nullptr || f.pos == nullptr || f.ids == nullptr || f.n_rows_dev == nullptr || f.S > 32 || f.h > V1_HMAX ||
        f.w_o == nullptr || f.attn16 == nullptr || f.dw_o == nullptr || f.db_o == nullptr) {
        set_error("fused_bwd16v1: needs the padding-skipping path (pos, ids, order lists), seq_len <= 32, w_o / attn16 / dw_o / db_o");
        return NRMS_EINVAL;
    }
    const long M = (long)f.n_seq * f.S;
    const V1Geom g = v1_geom(f.d, f.h);
    const Fused16V1BwdLayout L = v1_bwd_layout(M, f.n_seq, f.h);
    char* base = (char*)f.workspace;
    _Float16* btiles = (_Float16*)(base + L.btiles);
    _Float16* xtiles = (_Float16*)(base + L.xtiles);
    _Float16* qv16 = (_Float16*)(base + L.qv16);
    _Float16* dout16 = (_Float16*)(base + L.dout16);
    _Float16* dz16 = (_Float16*)(base + L.dz16);
    _Float16* dctx16 = (_Float16*)(base + L.dctx16);
    _Float16* dqkv16 = (_Float16*)(base + L.dqkv16);
    float* red_pool = (float*)(base + L.red_pool);
    float* red_attn = (float*)(base + L.red_attn);
    float* ebuf = (float*)(base + L.ebuf);
    Maps16V1 mp;
    mp.nmap_qkv = (int*)(base + L.maps);
    mp.nscale_qkv = (float*)(mp.nmap_qkv + 192 * V1_HMAX);
    mp.kmap_x = (int*)(mp.nscale_qkv + 192 * V1_HMAX);
    mp.nmap_add = mp.kmap_x + F16_KP;
    mp.nscale_add = (float*)(mp.nmap_add + F16_QP);
    mp.kmap_ctx = (int*)(mp.nscale_add + F16_QP);
    mp.nscale_o = (float*)(mp.kmap_ctx + F16_DP);
    mp.kmap_attn = (int*)(mp.nscale_o + F16_DP);
    float* sc = (float*)(base + L.scale);
    if (f.sc_out != nullptr) *f.sc_out = sc;
    { const int jr = fused_bwd16_join(stream); if (jr) return jr; }
    SideSet* ss = side_streams_for(stream);
    const bool side = ss != nullptr;
    hipStream_t s_add = side ? ss->s[0] : stream, s_qkv = side ? ss->s[1] : stream;
    SideJoinGuard join_guard(ss, stream);
    const int Mp = f.n_seq * 32;
    int rc;
    {
        Prep16bV1Args p{};
        p.g = g; p.q = f.q; p.w_qkv = f.w_qkv; p.b_qkv = f.b_qkv; p.w_o = f.w_o; p.w_add = f.w_add; p.q_vec = f.q_vec;
        p.btiles = btiles; p.xtiles = xtiles; p.qv16 = qv16;
        TimingScope ts("prep16", stream);
        hipLaunchKernelGGL(prep16bv1_kernel, dim3(2048), dim3(256), 0, stream, p);
        rc = launch_dout16((long)f.n_seq, f.d, g.hv, g.dkv, f.loss_scale, sc, f.dout, dout16, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(maps16v1_kernel, dim3(cdiv(192 * V1_HMAX, 256)), dim3(256), 0, stream, g, f.q, sc, mp);
        rc = check_launch("prep16bv1");
        if (rc) return rc;
    }
    const int n_groups = cdiv(f.n_seq, F16_WAVES) + 2;
    const int n_wg = n_groups < L.n_wg ? n_groups : L.n_wg;
    {
        // the pooling backward over the ten output blocks of W_O (the kernel of nrms_v0's encoders)
        Bwd16Args a{};
        a.n_seq = f.n_seq; a.S = f.S; a.d = f.d; a.h = g.hv; a.dk = g.dkv; a.q = f.q;
        a.n_groups = n_groups; a.n_cls = 3;
        a.x16 = (const _Float16*)f.x16; a.pos = f.pos; a.n_rows = f.n_rows_dev; a.ids = f.ids; a.order = f.order; a.order_cnt = f.order_cnt;
        a.btiles = btiles; a.bqkv32 = nullptr; a.qv16 = qv16;
        a.ctx16 = (const _Float16*)f.ctx16; a.t16 = (const _Float16*)f.t16; a.w = f.w; a.dout16 = dout16;
        a.dz16 = dz16; a.dctx16 = dctx16; a.dqkv16 = nullptr; a.red = red_pool; a.drop = f.drop;
        a.v1 = 1; a.dbg = 0;
        rc = launch_bwd16_pool(a, n_wg, false, stream);
        if (rc) return rc;
        TimingScope ts("red16", stream);
        hipLaunchKernelGGL(red16v1_kernel, dim3(cdiv(B16_RED, 32)), dim3(1024), 0, stream, 0, red_pool, n_wg, g, f.q, sc,
                           f.db_qkv, f.db_add, f.dq_vec, ebuf);
        rc = check_launch("red16v1");
        if (rc) return rc;
    }
    // helper stream 0: d(W_add) = dZ^T ctx, then d(W_O) | d(b_o) = d(o)^T [attn | 1] and the closed form of the all-padding titles
    if (side) { rc = side_order(ss, 0, stream, s_add, "fused_bwd16v1"); if (rc) return rc; }
    rc = launch_tn16(1, dz16, F16_QP, F16_QP, (const _Float16*)f.ctx16, F16_DP, F16_DP, Mp, nullptr, (float*)(base + L.p_add),
                     L.splits_add, mp.nmap_add, mp.kmap_ctx, mp.nscale_add, f.d, f.dw_add, s_add, "dwadd_bwd");
    if (rc) return rc;
    rc = launch_tn16(2, dctx16, F16_DP, F16_DP, (const _Float16*)f.attn16, F16_DP, F16_DP, Mp, nullptr, (float*)(base + L.p_o),
                     L.splits_o, mp.kmap_ctx, mp.kmap_attn, mp.nscale_o, f.d, f.dw_o, s_add, "dwo_bwd", f.db_o, f.order, f.order_cnt, f.n_seq);
    if (rc) return rc;
    {
        TimingScope ts("closed16", s_add);
        hipLaunchKernelGGL(closed16v1_wo_kernel, dim3(cdiv(f.d * f.d, 256)), dim3(256), 0, s_add, g, f.b_qkv, ebuf, f.dw_o, f.db_o);
        rc = check_launch("closed16v1_wo");
        if (rc) return rc;
    }
    if (side && hipEventRecord(ss->ev[2], s_add) != hipSuccess) { set_error("fused_bwd16v1: hipEventRecord failed"); return NRMS_ELAUNCH; }
    {
        Bwd16V1Args b{};
        b.n_seq = f.n_seq; b.S = f.S; b.g = g;
        b.x16 = (const _Float16*)f.x16; b.pos = f.pos; b.n_rows = f.n_rows_dev; b.ids = f.ids; b.order = f.order; b.order_cnt = f.order_cnt;
        b.btiles = btiles + (long)10 * 32 * F16_KP; b.dctx16 = dctx16; b.scratch = (_Float16*)(base + L.scratch);
        b.dqkv16 = dqkv16; b.red = red_attn;
        const size_t lds = (size_t)3 * F16_SLOT + (size_t)(V1_RED + 2 * F16_WAVES * 6 * 32) * 4;
        const hipError_t e = hipFuncSetAttribute((const void*)fused_bwd16v1_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("fused_bwd16v1: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
        {
            TimingScope ts("fused_bwd16_attn", stream);
            hipLaunchKernelGGL(fused_bwd16v1_attn_kernel, dim3(n_wg), dim3(F16_THREADS), lds, stream, b);
        }
        rc = check_launch("fused_bwd16v1_attn");
        if (rc) return rc;
        TimingScope ts("red16", stream);
        hipLaunchKernelGGL(red16v1_kernel, dim3(cdiv(192 * f.h, 32)), dim3(1024), 0, stream, 1, red_attn, n_wg, g, f.q, sc,
                           f.db_qkv, f.db_add, f.dq_vec, ebuf);
        hipLaunchKernelGGL(closed16v1_bv_kernel, dim3(cdiv(f.d, 32)), dim3(1024), 0, stream, g, f.w_o, ebuf, f.db_qkv);
        rc = check_launch("red16v1");
        if (rc) return rc;
    }
    // helper stream 1: d(W_qkv)[n][k] = sum_rows dQKV[r][n] x[r][k] (live rows), beside the dX GEMM
    if (side) { rc = side_order(ss, 1, stream, s_qkv, "fused_bwd16v1"); if (rc) return rc; }
    rc = launch_tn16(0, dqkv16, 192 * f.h, 192 * f.h, (const _Float16*)f.x16, F16_KP, F16_KP, (int)M, f.n_rows_dev, (float*)(base + L.p_qkv),
                     L.splits_qkv, mp.nmap_qkv, mp.kmap_x, mp.nscale_qkv, f.d, f.dw_qkv, s_qkv, "dwqkv_bwd");
    if (rc) return rc;
    if (side && hipEventRecord(ss->ev[3], s_qkv) != hipSuccess) { set_error("fused_bwd16v1: hipEventRecord failed"); return NRMS_ELAUNCH; }
    {
        Dx16Args gx{};
        gx.M = (int)M; gx.m_dev = f.n_rows_dev; gx.a16 = dqkv16; gx.xtiles = xtiles; gx.c = f.dx; gx.ldc = f.d; gx.d = f.d;
        gx.sc = sc; gx.lda = 192 * f.h; gx.n_slabs = 3 * f.h;
        rc = launch_dx16(gx, f.dx_fp16, stream);
    }
    if (side && rc == NRMS_OK) {
        if (f.defer_join) ss->pending = true;
        else if (hipStreamWaitEvent(stream, ss->ev[2], 0) != hipSuccess || hipStreamWaitEvent(stream, ss->ev[3], 0) != hipSuccess) {
            set_error("fused_bwd16v1: hipStreamWaitEvent failed");
            return NRMS_ELAUNCH;
        }
        join_guard.disarm();
    }
    return rc;
}

}  // namespace nrms
