// fp16 mode, backward: one wavefront per sequence, mirroring fused16.hip.
//
//   dout [n_seq, d] fp32 --(dout16_kernel: x loss scale, fp16, head-padded P16 order)--> fused_bwd16:
//     pooling backward (ds, dZ = ds q (1 - T^2)) -> dZ16 (for d(w_add)), column sums -> d(q_vec), d(b_add)
//     per head: d(ctx)^T = Wadd_h^T dZ^T + w (x) dout, dropout mask;  Q^T, K^T, V^T recomputed from x16;
//               P^T recomputed; dP^T = V d(ctx)^T; dS^T = P^T o (dP^T - delta);
//               dV^T = d(ctx)^T P, dQ^T = K^T dS^T, dK^T = Q^T dS  -> dqkv16 rows of the live tokens, d(b_qkv) sums
//   then  gemm16_dx:  dX = dQKV W'_qkv (live rows, fp32 out for the embedding scatter / the user encoder's input gradient)
//         gemm16_tn:  d(W_qkv) = dQKV^T X,  d(W_add) = dZ^T ctx   (contraction over token rows: ds_read_b64_tr_b16 operands)
//
// Every product whose contraction index sits in the REGISTERS of a 32x32 accumulator takes that accumulator as an
// operand directly (acc_frag); where the contraction index sits on the LANES, the tile is first transposed by one
// extra product with a (k-permuted) identity, which costs 2 MFMAs and no LDS traffic.
// Replaces autograd through model/nrms_v0.py:13-23,46-76,100-126,154-176,188-199 (`loss.backward()`, train_eval.py:126).
#include <stdlib.h>

#include "fused16_bwd.h"

namespace nrms {

// SB = 32-row blocks per sequence (1: titles / short histories, two workgroups per CU; 2: sequences of up to 64 rows, one
// workgroup per CU with up to 512 registers per lane) -- see fused16.hip.
template <int SB>
__global__ __launch_bounds__(F16_THREADS, SB == 1 ? 8 / F16_WAVES : 1) void fused_bwd16_pool_kernel(Bwd16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + 3 * F16_SLOT);         // [B16_RED] column sums of this workgroup
    // Column sums are collected WITHOUT atomics: every wave writes its share to a row of its own (stg_t: the 2 x 7 tile
    // sums of the pooling backward per row block; stg_h: an all-padding title's d(b_v) share of one head, double buffered
    // over the head loop's barriers) and one thread per column adds the waves' shares in ascending wave order -- the same
    // bits on every run (a deterministic option for gradient parity).
    float* stg_t = red + B16_RED;                                       // [F16_WAVES][SB][2 F16_QT][32]
    float* stg_h = stg_t + F16_WAVES * SB * 2 * F16_QT * 32;            // [2][F16_WAVES][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int S = a.S;
    constexpr int DP = F16_DP, QP = F16_QP;

    for (int i = tid; i < B16_RED; i += F16_THREADS) red[i] = 0.f;

    // constant operands: transposition identity (k-permuted) and the column selectors of the token reductions
    h8 idf[2], sel[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            idf[s][j] = (_Float16)(l32 == 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3) ? 1.0f : 0.0f);
            sel[s][j] = (_Float16)(l32 == 16 * s + 8 * hh + j ? 1.0f : 0.0f);
        }

    // title lists: [0] titles with a real token (n_cls 3: the long ones), [1] all-padding titles, [2] (n_cls 3) short titles
    int n_ne = a.n_seq, n_e = 0, n_sh = 0, g_ne = a.n_groups, g_sh = 0;
    if (a.order != nullptr) {
        n_ne = a.order_cnt[0];
        n_e = a.order_cnt[1];
        g_ne = (n_ne + F16_WAVES - 1) / F16_WAVES;
        if (a.n_cls == 3) { n_sh = a.order_cnt[2]; g_sh = (n_sh + F16_WAVES - 1) / F16_WAVES; }
    }
    TileRing ring;
    ring.smem = smem; ring.src = a.btiles; ring.n_tiles = a.h; ring.tid = tid; ring.l32 = l32; ring.hh = hh; ring.dbg = 0;

#pragma unroll 1
    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        // ---- which sequence this wave owns in this group (groups: long / short / all-padding titles, four per group)
        int slot_id = grp * F16_WAVES + wave;
        bool valid;
        int seq;
        if (grp < g_ne) { valid = slot_id < n_ne; seq = valid ? (a.order != nullptr ? a.order[slot_id] : slot_id) : 0; }
        else if (grp < g_ne + g_sh) {
            slot_id -= g_ne * F16_WAVES;
            valid = slot_id < n_sh;
            seq = valid ? a.order[2 * (long)a.n_seq + slot_id] : 0;
        } else {
            slot_id -= (g_ne + g_sh) * F16_WAVES;
            valid = slot_id < n_e;
            seq = valid ? a.order[a.n_seq + slot_id] : 0;
        }
        const long tok0 = (long)seq * S;
        bool empty = false;
        if (SB == 1 && a.ids != nullptr && valid) {
            const bool is_pad = lane < S ? a.ids[tok0 + lane] == 0 : true;
            empty = __ballot(is_pad) == ~0ull;
        }
        const bool live = valid && !empty;
        // a short title (n_cls 3, the third list): its n real tokens are a prefix, and the attention kernel processes it in n + 1
        // tile rows -- one per real token, one for all its padding tokens.  Its d(ctx) leaves this kernel in that form: column
        // c < n = token c, column n = the SUM over the padding tokens' rows (they share one context row in the forward, so their
        // gradients add), by one product with the selector SelT[tok][col] = (min(tok, n) == col).
        int nshort = -1;
        h8 selT[2] = {h8{0, 0, 0, 0, 0, 0, 0, 0}, h8{0, 0, 0, 0, 0, 0, 0, 0}};
        if (SB == 1 && a.n_cls == 3 && live && grp >= g_ne) {
            nshort = __popcll(__ballot(lane < S && a.ids[tok0 + lane] != 0));
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int tok = crow32(8 * s + j, hh);                 // the k slot's token (the order of acc_frag)
                    const int src = tok < nshort ? tok : (tok < S ? nshort : -1);
                    selT[s][j] = (_Float16)(src == l32 ? 1.0f : 0.0f);
                }
        }
        // tile stream of this kernel: the h tiles Wadd_h^T (the first h tiles of btiles)
        ring.load(0);                                                   // lands while the pooling backward runs

        bool tok_ok[SB];
        float wgt[SB], dw[SB];
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            tok_ok[b] = valid && 32 * b + l32 < S;
            wgt[b] = tok_ok[b] ? a.w[tok0 + 32 * b + l32] : 0.f;
        }

        // ================= pooling backward =================
        // dw_tok = <dout, ctx_tok>: A = dout (the same row in every lane), B = ctx fragments; all rows of the result equal
        float aw = 0.f;
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            f32x16 acc = zero16();
            const _Float16* dsrc = a.dout16 + (long)seq * DP + 8 * hh;
            const _Float16* csrc = a.ctx16 + frag_off((long)seq * SB + b, F16_CS, 0, l32, hh);
#pragma unroll
            for (int g = 0; g < F16_CS / 4; ++g) {
                h8 da[4], cb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    da[i] = *reinterpret_cast<const h8*>(dsrc + 16 * (4 * g + i));
                    cb[i] = *reinterpret_cast<const h8*>(csrc + 512 * (4 * g + i));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = mfma32h(da[i], cb[i], acc);
            }
            dw[b] = acc[0];
            aw += wgt[b] * dw[b];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) aw += __shfl_xor(aw, o, 64);

        // dZ[tok][q] = ds q_vec[q] (1 - T^2),  U[tok][q] = ds T  (column sums of U = d(q_vec), of dZ = d(b_add)).
        // The dZ fragments stay in registers: they are the B operand of every head's d(ctx) product below.
        h8 zf[SB][16];
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            const float ds = wgt[b] * (dw[b] - aw);                      // 0 for lanes beyond the sequence (wgt = 0)
            const _Float16* tsrc = a.t16 + frag_off((long)seq * SB + b, QP / 16, 0, l32, hh);
#pragma unroll
            for (int s = 0; s < 14; ++s) zf[b][s] = *reinterpret_cast<const h8*>(tsrc + 512 * s);    // tanh(.) first, dZ in place
#pragma unroll
            for (int t = 0; t < F16_QT; ++t) {
                f32x16 accz = zero16(), accu = zero16();
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int s = 2 * t + s2;
                    const h8 qf = *reinterpret_cast<const h8*>(a.qv16 + 16 * s + 8 * hh);
                    h8 dz, u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float tv = (float)zf[b][s][j];
                        dz[j] = (_Float16)(ds * (float)qf[j] * (1.0f - tv * tv));
                        u[j] = (_Float16)(ds * tv);
                    }
                    zf[b][s] = dz;
                    // (zero beyond the sequence: ds = 0 there)
                    if (valid && !F16_DBG(a.dbg, 4)) *reinterpret_cast<h8*>(a.dz16 + frag_off((long)seq * SB + b, QP / 16, s, l32, hh)) = dz;
                    // transposing products: D[tok][n] = X[tok][q = 32 t + n]  (rows = tokens in registers)
                    accz = mfma32h(dz, sel[s2], accz);
                    accu = mfma32h(u, sel[s2], accu);
                }
                // this wave's share of d(b_add) (columns of dZ) and d(q_vec) (columns of ds T): both lane halves hold
                // partial sums over different token rows
                float cz = valid ? regsum(accz) : 0.f, cu = valid ? regsum(accu) : 0.f;
                cz += __shfl_xor(cz, 32, 64);
                cu += __shfl_xor(cu, 32, 64);
                if (hh == 0) {
                    float* row = stg_t + ((wave * SB + b) * 2 * F16_QT + 2 * t) * 32 + l32;
                    row[0] = cz;
                    row[32] = cu;
                }
            }
            zf[b][14] = h8{0, 0, 0, 0, 0, 0, 0, 0};
            zf[b][15] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        ring.store(0);
        ring.load(1);
        ring.store(1);
        const bool any_closed = __syncthreads_or((!live && valid) ? 1 : 0) != 0;     // (also the barrier the ring needs here)
        for (int i = tid; i < 2 * F16_QT * 32; i += F16_THREADS) {           // fixed order: waves ascending, row blocks ascending
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < F16_WAVES; ++w)
#pragma unroll
                for (int b = 0; b < SB; ++b) sum += stg_t[((w * SB + b) * 2 * F16_QT) * 32 + i];
            const int r = i >> 5, c = i & 31;                                // r = 2 t + {0: dZ, 1: ds T}
            red[B16_RED_QKV + (r & 1) * QP + 32 * (r >> 1) + c] += sum;
        }

        // ================= d(ctx)^T per head -> dctx16 =================
        // d(ctx)^T[f][tok] = sum_q Wadd[q][f] dZ[tok][q] + w_tok dout[f], then the forward's dropout mask
        int n = 0;
#pragma unroll 1
        for (int head = 0; head < a.h; ++head) {
            ring.load(n + 2);
            const h8 d0 = *reinterpret_cast<const h8*>(a.dout16 + (long)seq * DP + head * 32 + 8 * hh);
            const h8 d1 = *reinterpret_cast<const h8*>(a.dout16 + (long)seq * DP + head * 32 + 16 + 8 * hh);
            float closed = 0.f;                                           // an all-padding title's share of d(b_v) of this head
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                f32x16 dct = zero16();
                if (!F16_DBG(a.dbg, 8)) tile_mma<true>(dct, ring, n, zf[b]);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    dct[r] += wgt[b] * (float)d0[r];
                    dct[8 + r] += wgt[b] * (float)d1[r];
                }
                if (a.drop.thresh != 0u) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {               // the forward's mask: same two calls per lane
                        const uint64_t e0 = (uint64_t)(tok0 + 32 * b + l32) * (uint64_t)DP + (uint64_t)(head * 32 + 16 * c + 8 * hh);
                        float sc[8];
                        dropout_scale8(a.drop.seed, 1u, e0 >> 3, a.drop.thresh16, a.drop.inv_keep, sc);
#pragma unroll
                        for (int e = 0; e < 8; ++e) dct[8 * c + e] *= sc[e];
                    }
                }
                if (a.v1 && live && nshort >= 0 && l32 >= 16) {
                    _Float16* dcrow = a.dctx16 + frag_off((long)seq * SB + b, F16_CS, 2 * head, l32, hh);
                    *reinterpret_cast<h8*>(dcrow) = h8{0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<h8*>(dcrow + 512) = h8{0, 0, 0, 0, 0, 0, 0, 0};
                }
                if (live && !F16_DBG(a.dbg, 4)) {                     // (zero beyond the sequence: dZ and w are 0 there); the
                    // attention kernel reads the rows of titles with a real token only
                    _Float16* dcrow = a.dctx16 + frag_off((long)seq * SB + b, F16_CS, 2 * head, l32, hh);
                    if (SB == 1 && nshort >= 0) {
                        const f32x16 xt = transpose32(dct, idf);                       // [tok][f]
                        f32x16 dcp = mfma32h(acc_frag(xt, 0), selT[0], zero16());       // [f][col] = sum_tok xt[tok][f] SelT[tok][col]
                        dcp = mfma32h(acc_frag(xt, 1), selT[1], dcp);
                        if (l32 < 16) {                                                 // columns 0 .. n carry data, n + 1 .. 15 zeros
                            *reinterpret_cast<h8*>(dcrow) = acc_frag(dcp, 0);
                            *reinterpret_cast<h8*>(dcrow + 512) = acc_frag(dcp, 1);
                        }
                    } else {
                        *reinterpret_cast<h8*>(dcrow) = acc_frag(dct, 0);
                        *reinterpret_cast<h8*>(dcrow + 512) = acc_frag(dct, 1);
                    }
                }
                if (!live && valid) {
                    // closed form (all-padding title): dS = 0, every dV row = mean of d(ctx) rows => d(b_v) += sum_tok d(ctx)
                    closed += regsum(transpose32(dct, idf));
                }
            }
            if (any_closed) {
                closed += __shfl_xor(closed, 32, 64);
                if (hh == 0) stg_h[((head & 1) * F16_WAVES + wave) * 32 + l32] = closed;
            }
            ring.store(n + 2);
            __syncthreads();
            if (any_closed && tid < 32) {                                 // (the buffer of this parity is rewritten two barriers later)
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < F16_WAVES; ++w) sum += stg_h[((head & 1) * F16_WAVES + w) * 32 + tid];
                red[(3 * head + 2) * 32 + tid] += sum;
            }
            ++n;
        }
        __syncthreads();                                          // the ring restarts: nobody may still read a slot
    }
    __syncthreads();
    float* out = a.red + (long)blockIdx.x * B16_RED;
    for (int i = tid; i < B16_RED; i += F16_THREADS) out[i] = red[i];
}

// The attention part of the backward, on the sequences with a real token only.  With a 3-class title list (n_cls 3, SB = 1)
// the groups are [pairs of short titles][long titles]: a pair shares one 32-row tile exactly as in the forward
// (fused16.hip, fused_fwd16p_kernel) -- rows 16 p .. 16 p + n_p - 1 the real tokens of title p, row 16 p + n_p all its padding
// tokens (as a key: logit + log multiplicity; as a query: the sum of their d(ctx) rows, which the pooling kernel has already
// formed), cross-title entries of S^T masked by the additive bias -- so the recomputed Q | K | V tiles, the softmax and the
// five gradient products serve two titles at once.  The shared row's dQ | dK | dV are the sums over the padding tokens: they
// enter the bias gradients (column sums over the tile's rows) and are not stored (a padding token has no dX, no share of dW).
template <int SB>
__global__ __launch_bounds__(F16_THREADS, SB == 1 ? 8 / F16_WAVES : 1) void fused_bwd16_attn_kernel(Bwd16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + 3 * F16_SLOT);
    float* stg = red + B16_RED;                       // [2][F16_WAVES][3][32]: per-wave shares of d(b_q), d(b_k), d(b_v) of one head,
                                                      // double buffered over the head loop's barriers (no atomics: see the pool kernel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int S = a.S;
    constexpr int KP = F16_KP;
    constexpr float NEG = -3.0e38f;
    for (int i = tid; i < B16_RED; i += F16_THREADS) red[i] = 0.f;
    h8 idf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) idf[s][j] = (_Float16)(l32 == 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3) ? 1.0f : 0.0f);
    int n_ne = a.n_seq, g_ne = a.n_groups, n_sh = 0, g_pair = 0;
    if (a.order != nullptr) {
        n_ne = a.order_cnt[0];
        g_ne = (n_ne + F16_WAVES - 1) / F16_WAVES;
        if (SB == 1 && a.n_cls == 3) { n_sh = a.order_cnt[2]; g_pair = (n_sh + 2 * F16_WAVES - 1) / (2 * F16_WAVES); }
    }
    TileRing ring;
    ring.smem = smem; ring.src = a.btiles + (long)a.h * 32 * KP; ring.n_tiles = 3 * a.h; ring.tid = tid; ring.l32 = l32; ring.hh = hh; ring.dbg = 0;
    const h8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
#pragma unroll 1
    for (int grp = blockIdx.x; grp < g_pair + g_ne; grp += gridDim.x) {
        const bool pair = SB == 1 && grp < g_pair;                      // uniform over the workgroup
        // ---- the titles of this wave: two short ones (pair) or one
        int seq2[2] = {0, 0};
        bool val2[2] = {false, false};
        int nl2[2] = {0, 0};                                            // real tokens of a short title
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
            val2[0] = slot_id < n_ne;
            seq2[0] = val2[0] ? (a.order != nullptr ? a.order[slot_id] : slot_id) : 0;
        }
        const bool valid = val2[0];
        const int seq = seq2[0];
        const long tok0 = (long)seq * S;
        bool empty = false;
        if (SB == 1 && !pair && a.ids != nullptr && valid) {
            const bool is_pad = lane < S ? a.ids[tok0 + lane] == 0 : true;
            empty = __ballot(is_pad) == ~0ull;
        }
        const bool live = pair ? (val2[0] || val2[1]) : (valid && !empty);
        // this lane's tile row in pair mode: title myp, row myr of its 16
        const int myp = pair ? (l32 >> 4) : 0, myr = pair ? (l32 & 15) : l32;
        const int myn = myp ? nl2[1] : nl2[0];
        const bool myval = myp ? val2[1] : val2[0];
        const int myseq = myp ? seq2[1] : seq2[0];
        f32x16 kbias;                                                   // SB = 1: additive softmax mask of the key rows this lane holds
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
        bool tok_ok[SB];
        long drow[SB];                                                  // x16 / dqkv16 row of this lane's token, -1: none
        h8 xf[SB][F16_KS];
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            long xrow;
            if (SB == 1 && pair) {
                tok_ok[b] = myval && myr <= myn;
                drow[b] = (myval && myr < myn) ? (long)a.pos[(long)myseq * S + myr] : -1;     // real tokens only
                xrow = drow[b] >= 0 ? drow[b] : ((myval && myr == myn && myn < S) ? (long)*a.n_rows : -1);
            } else {
                tok_ok[b] = valid && 32 * b + l32 < S;
                drow[b] = tok_ok[b] && live ? (a.pos != nullptr ? (long)a.pos[tok0 + 32 * b + l32] : tok0 + 32 * b + l32) : -1;
                // a padding token inside a live title reads the pad row (zeros + the ones column => Q|K|V = bias)
                xrow = (tok_ok[b] && live && drow[b] < 0) ? (long)*a.n_rows : drow[b];
            }
            const _Float16* xr = a.x16 + (xrow < 0 ? 0 : xrow) * KP + 8 * hh;
#pragma unroll
            for (int s = 0; s < F16_KS; ++s) xf[b][s] = *reinterpret_cast<const h8*>(xr + 16 * s);
            if (xrow < 0) {
#pragma unroll
                for (int s = 0; s < F16_KS; ++s) xf[b][s] = z8;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int head = 0; head < a.h; ++head) {
            // d(ctx)^T of this head as operand fragments (rows f = k); zero beyond the sequence
            h8 dc[SB][2];
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                // pair: column myr of title myp's COMPRESSED d(ctx) (fused_bwd16_pool_kernel); a missing second title: zeros
                const _Float16* dcsrc = (SB == 1 && pair) ? a.dctx16 + frag_off((long)myseq, F16_CS, 2 * head, myr, hh)
                                                          : a.dctx16 + frag_off((long)seq * SB + b, F16_CS, 2 * head, l32, hh);
                dc[b][0] = *reinterpret_cast<const h8*>(dcsrc);             // (zeros beyond the sequence)
                dc[b][1] = *reinterpret_cast<const h8*>(dcsrc + 512);
                if (SB == 1 && pair && !myval) { dc[b][0] = z8; dc[b][1] = z8; }
            }
            // ---- tiles W'_q, W_k, W_v: Q^T, K^T, V^T [f][tok], kept as operand fragments only (bias: the ones column)
            h8 qf[SB][2], kf[SB][2], vf[SB][2];
            ring.load(n + 2);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                f32x16 t = zero16();
                if (live) tile_mma<true>(t, ring, n, xf[b]);
                qf[b][0] = acc_frag(t, 0); qf[b][1] = acc_frag(t, 1);
            }
            ring.store(n + 2);
            __syncthreads();
            ++n;
            ring.load(n + 2);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                f32x16 t = zero16();
                if (live) tile_mma<true>(t, ring, n, xf[b]);
                kf[b][0] = acc_frag(t, 0); kf[b][1] = acc_frag(t, 1);
            }
            ring.store(n + 2);
            __syncthreads();
            ++n;
            ring.load(n + 2);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                f32x16 t = zero16();
                if (live) tile_mma<true>(t, ring, n, xf[b]);
                vf[b][0] = acc_frag(t, 0); vf[b][1] = acc_frag(t, 1);
            }
            ring.store(n + 2);
            // column sums of dQ', dK, dV (= of d(ctx)) over this title's rows go to this wave's staging row as soon as they
            // are complete (short live ranges: the kernel sits at the 256-register limit); a wave without a live title adds zeros
            float* srow = stg + (((head & 1) * F16_WAVES + wave) * 3) * 32 + l32;
            if (!live && hh == 0) { srow[0] = 0.f; srow[32] = 0.f; srow[64] = 0.f; }
            if (live) {
                // ---- per query block ib: P^T[jb][ib] (rows = keys of block jb, columns = queries), dS^T[jb][ib]
                h8 pf[SB][SB][2], sf[SB][SB][2];                         // [jb][ib] operand fragments
#pragma unroll
                for (int ib = 0; ib < SB; ++ib) {
                    f32x16 pt[SB];
                    float m = -3.0e38f;
#pragma unroll
                    for (int jb = 0; jb < SB; ++jb) {
                        pt[jb] = mfma32h(kf[jb][0], qf[ib][0], zero16());
                        pt[jb] = mfma32h(kf[jb][1], qf[ib][1], pt[jb]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (SB == 1) pt[jb][r] += kbias[r];           // rows beyond the sequence / the other title; padding multiplicity
                            else pt[jb][r] = 32 * jb + crow32(r, hh) < S ? pt[jb][r] : NEG;
                            m = fmaxf(m, pt[jb][r]);
                        }
                    }
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    float sum = 0.f;
#pragma unroll
                    for (int jb = 0; jb < SB; ++jb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float p = (SB == 1 || 32 * jb + crow32(r, hh) < S) ? __expf(pt[jb][r] - m) : 0.f;
                            pt[jb][r] = p;
                            sum += p;
                        }
                    sum += __shfl_xor(sum, 32, 64);
                    const float inv = 1.0f / sum;
                    // dP^T[j][i] = sum_f V^T[f][j] d(ctx)^T[f][i];  dS^T = P^T o (dP^T - delta_i)
                    f32x16 dst[SB];
                    float delta = 0.f;
#pragma unroll
                    for (int jb = 0; jb < SB; ++jb) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) pt[jb][r] *= inv;
                        dst[jb] = mfma32h(vf[jb][0], dc[ib][0], zero16());
                        dst[jb] = mfma32h(vf[jb][1], dc[ib][1], dst[jb]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) delta += pt[jb][r] * dst[jb][r];
                    }
                    delta += __shfl_xor(delta, 32, 64);
#pragma unroll
                    for (int jb = 0; jb < SB; ++jb) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[jb][r] = pt[jb][r] * (dst[jb][r] - delta);
                        pf[jb][ib][0] = acc_frag(pt[jb], 0); pf[jb][ib][1] = acc_frag(pt[jb], 1);
                        sf[jb][ib][0] = acc_frag(dst[jb], 0); sf[jb][ib][1] = acc_frag(dst[jb], 1);
                    }
                }
                // ---- d(ctx) [i][f] per query block; d(b_v) = sum_i d(ctx)_i (rows of P sum to 1)
                h8 dx0[SB][2];
                {
                    float sv = 0.f;
#pragma unroll
                    for (int ib = 0; ib < SB; ++ib) {
                        const f32x16 dctx = transpose_frags(dc[ib][0], dc[ib][1], idf);
                        sv += regsum(dctx);
                        dx0[ib][0] = acc_frag(dctx, 0); dx0[ib][1] = acc_frag(dctx, 1);
                    }
                    sv += __shfl_xor(sv, 32, 64);
                    if (hh == 0) srow[64] = sv;
                }
                float sk = 0.f;
#pragma unroll
                for (int jb = 0; jb < SB; ++jb) {
                    _Float16* orow = a.dqkv16 + (drow[jb] < 0 ? 0 : drow[jb]) * (long)B16_DQ + head * 96 + 8 * hh;
                    // ---- dV^T[f][j] = sum_i d(ctx)[i][f] P[i][j]   (j in block jb)
                    {
                        f32x16 dv = zero16();
#pragma unroll
                        for (int ib = 0; ib < SB; ++ib) {
                            const f32x16 p = transpose_frags(pf[jb][ib][0], pf[jb][ib][1], idf);        // [i][j]
                            dv = mfma32h(dx0[ib][0], acc_frag(p, 0), dv);
                            dv = mfma32h(dx0[ib][1], acc_frag(p, 1), dv);
                        }
                        if (drow[jb] >= 0) {
                            *reinterpret_cast<h8*>(orow + 64) = acc_frag(dv, 0);
                            *reinterpret_cast<h8*>(orow + 64 + 16) = acc_frag(dv, 1);
                        }
                    }
                    // ---- dK^T[f][j] = sum_i Q'[i][f] dS[i][j]   (j in block jb)
                    {
                        f32x16 dk = zero16();
#pragma unroll
                        for (int ib = 0; ib < SB; ++ib) {
                            const f32x16 q = transpose_frags(qf[ib][0], qf[ib][1], idf);                // [i][f]
                            const f32x16 dsn = transpose_frags(sf[jb][ib][0], sf[jb][ib][1], idf);      // [i][j]
                            dk = mfma32h(acc_frag(q, 0), acc_frag(dsn, 0), dk);
                            dk = mfma32h(acc_frag(q, 1), acc_frag(dsn, 1), dk);
                        }
                        const h8 dk0 = acc_frag(dk, 0), dk1 = acc_frag(dk, 1);
                        if (drow[jb] >= 0) {
                            *reinterpret_cast<h8*>(orow + 32) = dk0;
                            *reinterpret_cast<h8*>(orow + 32 + 16) = dk1;
                        }
                        sk += regsum(transpose_frags(dk0, dk1, idf));
                    }
                }
                sk += __shfl_xor(sk, 32, 64);
                if (hh == 0) srow[32] = sk;
                // ---- dQ'^T[f][i] = sum_j K[j][f] dS^T[j][i]   (i in block ib)
                float sq = 0.f;
#pragma unroll
                for (int ib = 0; ib < SB; ++ib) {
                    f32x16 dq = zero16();
#pragma unroll
                    for (int jb = 0; jb < SB; ++jb) {
                        const f32x16 k = transpose_frags(kf[jb][0], kf[jb][1], idf);                    // [j][f]
                        dq = mfma32h(acc_frag(k, 0), sf[jb][ib][0], dq);
                        dq = mfma32h(acc_frag(k, 1), sf[jb][ib][1], dq);
                    }
                    const h8 dq0 = acc_frag(dq, 0), dq1 = acc_frag(dq, 1);
                    if (drow[ib] >= 0) {
                        _Float16* orow = a.dqkv16 + drow[ib] * (long)B16_DQ + head * 96 + 8 * hh;
                        *reinterpret_cast<h8*>(orow) = dq0;
                        *reinterpret_cast<h8*>(orow + 16) = dq1;
                    }
                    sq += regsum(transpose_frags(dq0, dq1, idf));
                }
                sq += __shfl_xor(sq, 32, 64);
                if (hh == 0) srow[0] = sq;
            }
            __syncthreads();
            if (tid < 96) {                                               // fixed order: waves ascending
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < F16_WAVES; ++w) sum += stg[(((head & 1) * F16_WAVES + w) * 3) * 32 + tid];
                red[3 * head * 32 + tid] += sum;
            }
            ++n;
        }
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            if (drow[b] >= 0) {                                   // heads the model does not have: zero columns (dX contracts over all 960)
                for (int head = a.h; head < 10; ++head) {
                    _Float16* orow = a.dqkv16 + drow[b] * (long)B16_DQ + head * 96 + 8 * hh;
#pragma unroll
                    for (int c = 0; c < 96; c += 16) *reinterpret_cast<h8*>(orow + c) = z8;
                }
            }
        }
        __syncthreads();                                          // the ring restarts: nobody may still read a slot
    }
    __syncthreads();
    float* out = a.red + (long)blockIdx.x * B16_RED;
    for (int i = tid; i < B16_RED; i += F16_THREADS) out[i] = red[i];
}

// ---- loss scale of the fp16 backward, chosen ON THE DEVICE from the upstream gradient it is about to round to fp16:
// the power of two that puts max |dout| into [64, 128) (fp16 keeps 11 bits down to 6e-5 and overflows above 65504, so the
// gradient tensors derived from dout have ~2^9 of head room and ~2^20 below them whatever the caller's loss reduction,
// batch size or data-parallel world size is).  A fixed scale (desc.loss_scale > 0) is used as given.
// sc[0] = scale, sc[1] = 1 / scale (both exact powers of two); every kernel that removes the scale reads sc[1].
// desc.loss_scale = -n (a negative integer): n more powers of two of head room, max |dout| into [64, 128) / 2^n -- the
// caller's back-off after nrms_adam_step_guarded / nrms_grad_guard reported non-finite gradients.
__device__ __forceinline__ float loss_scale_from_max(unsigned max_bits, float fixed) {
    if (fixed > 0.f) return fixed;
    const int e = (int)((max_bits >> 23) & 0xFF) - 127;             // 2^e <= max < 2^(e+1)   (max_bits: |x| as uint)
    if (max_bits == 0u || e == 128) return 1.0f;                    // all zero, or inf / nan: nothing sensible to scale (the
                                                                    // gradients come out non-finite and the guarded optimizer skips them)
    int k = 6 - e + (int)fmaxf(fixed, -24.f);
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    return __uint_as_float((unsigned)(k + 127) << 23);
}

__global__ __launch_bounds__(256) void absmax_kernel(long n, const float* x, unsigned* max_bits) {
    unsigned m = 0u;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        const long n4 = n / 4;
        long i = tid;
        for (; i + 3 * nth < n4; i += 4 * nth) {               // four loads in flight per thread (one was a chain of HBM round trips)
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = x4[i + u * nth];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[u][e]) & 0x7FFFFFFFu);   // nan / inf sort above every finite value
        }
        for (; i < n4; i += nth) {
            const f32x4 v = x4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[e]) & 0x7FFFFFFFu);
        }
        for (long i = (n & ~3L) + tid; i < n; i += nth) m = max(m, __float_as_uint(x[i]) & 0x7FFFFFFFu);
    } else {
        for (long i = tid; i < n; i += nth) m = max(m, __float_as_uint(x[i]) & 0x7FFFFFFFu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    // one atomic per WORKGROUP: 4 096 same-address atomics (one per wave) serialised at the L2 for 45 of this kernel's 54 us
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m != 0u) atomicMax(max_bits, m);
    }
}

// dout16[n][P16(padded f)] = fp16(dout[n][f] * scale), zero in the padding columns
__global__ __launch_bounds__(256) void dout16_kernel(long n_seq, int d, int h, int dk, float fixed_scale, const unsigned* max_bits,
                                                     float* sc, const float* dout, _Float16* dout16) {
    const float scale = loss_scale_from_max(*max_bits, fixed_scale);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[0] = scale; sc[1] = 1.0f / scale; }
    const long total = n_seq * F16_DP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / F16_DP;
        const int p = (int)(i - r * F16_DP);
        const int b16 = p >> 4, t = p & 15, hh = t >> 3, jj = t & 7;
        const int fpad = 16 * b16 + 8 * (jj >> 2) + 4 * hh + (jj & 3);
        const int head = fpad >> 5, f = fpad & 31;
        float v = 0.f;
        if (head < h && f < dk) v = dout[r * d + head * dk + f] * scale;
        dout16[i] = (_Float16)v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward weight planes: tile stream of the fused kernel, the dX weight tiles, q_vec in fp16
struct Prep16bArgs {
    int d, h, dk, q;
    const float* w_qkv;   // [3d][d]
    const float* b_qkv;   // [3d]
    const float* w_add;   // [q][d]
    const float* q_vec;   // [q]
    float* bqkv32;        // [3h][32] (Q part pre-scaled)
    _Float16* btiles;     // [4h][32][KP]
    _Float16* xtiles;     // dX GEMM (gemm16_dx_kernel): [DX_SLABS][4 k-steps][10 n-tiles][lane half][32 n][8 k]: the operand
                          // fragments of W' (rows = dqkv16 columns, 64 per slab; columns = input features) in the order a
                          // wave reads them -- one contiguous KiB per (k-step, n-tile), filled by LDS-DMA
    _Float16* qv16;       // [QP]
};

__global__ __launch_bounds__(256) void prep16b_kernel(Prep16bArgs a) {
    const float qscale = 1.0f / sqrtf((float)a.dk);
    constexpr int KP = F16_KP;
    const long n1 = (long)4 * a.h * 32 * KP, n2 = (long)30 * 32 * KP, n3 = F16_QP, n4 = (long)3 * a.h * 32;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2 + n3 + n4; i += (long)gridDim.x * blockDim.x) {
        if (i < n1) {
            const int c = (int)(i % KP);
            const long r = i / KP;
            const int f = (int)(r & 31), tile = (int)(r >> 5);
            float v = 0.f;
            if (tile < a.h) {                                   // Wadd_h^T: row f, column q (natural order)
                const int head = tile;
                if (f < a.dk && c < a.q) v = a.w_add[(long)c * a.d + head * a.dk + f];
            } else {
                const int t3 = tile - a.h, head = t3 / 3, which = t3 - 3 * head;
                if (f < a.dk && c < a.d)
                    v = a.w_qkv[((long)which * a.d + head * a.dk + f) * a.d + c] * (which == 0 ? qscale : 1.0f);
                if (f < a.dk && c == a.d) v = a.b_qkv[which * a.d + head * a.dk + f] * (which == 0 ? qscale : 1.0f);   // ones column
            }
            a.btiles[i] = (_Float16)v;
        } else if (i < n1 + n2) {
            const long j = i - n1;
            // position j = ((((slab * 4 + s) * 10 + t) * 2 + hh) * 32 + l) * 8 + e  ->  W'[m][n], m = 64 slab + 16 s + 8 hh + e, n = 32 t + l
            const int e = (int)(j & 7), l = (int)((j >> 3) & 31), hh2 = (int)((j >> 8) & 1);
            long r = j >> 9;
            const int t = (int)(r % 10); r /= 10;
            const int s4 = (int)(r & 3), slab = (int)(r >> 2);
            const int m = 64 * slab + 16 * s4 + 8 * hh2 + e;     // dqkv16 column
            const int k = 32 * t + l;                            // input feature (output column of dX)
            const int head = m / 96, rem = m - head * 96, which = rem >> 5, p = rem & 31;
            const int s = p >> 4, tt = p & 15, hh = tt >> 3, jj = tt & 7;
            const int f = 16 * s + 8 * (jj >> 2) + 4 * hh + (jj & 3);       // feature held at memory position p (acc_frag order)
            float v = 0.f;
            if (head < a.h && f < a.dk && k < a.d)
                v = a.w_qkv[((long)which * a.d + head * a.dk + f) * a.d + k] * (which == 0 ? qscale : 1.0f);
            a.xtiles[j] = (_Float16)v;
        } else if (i < n1 + n2 + n3) {
            const long j = i - n1 - n2;
            a.qv16[j] = (_Float16)(j < a.q ? a.q_vec[j] : 0.f);
        } else {
            const long j = i - n1 - n2 - n3;
            const int f = (int)(j & 31), tile = (int)(j >> 5), head = tile / 3, which = tile - 3 * head;
            a.bqkv32[j] = f < a.dk ? a.b_qkv[which * a.d + head * a.dk + f] * (which == 0 ? qscale : 1.0f) : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dX[r][k] = inv_scale * sum_m dqkv16[r][m] W'[m][k]  for rows r < *n_rows (compact live rows) -- fp32 [rows][ldc].
// A plain LDS-staged GEMM, sized by what bounds it: 0.56 GB of dQKV16 in + 0.35 GB of dX out against 180 GFLOP, i.e. HBM
// (0.18 ms at 5 TB/s) well before the matrix pipe (0.08 ms).  256 rows x all 320 columns per workgroup, 8 waves as
// 4 (rows) x 2 (columns): a wave owns 64 x 160 = 2 x 5 accumulator tiles, so one 64-deep slab costs it 40 MFMAs for 28 LDS
// fragment reads.  Both operands reach LDS by LDS-DMA (no staging registers, no ds_write pass, no ordinary load in the loop --
// hipcc drains the DMA queue in front of the first use of any ordinary load):
//   A slab [256 rows][64 k] row-major, 128 B per row, filled in whole 128-B lines (8 rows per DMA instruction); the 16-byte
//     pieces of a row are XOR-swizzled by (row >> 1) & 7 -- on the SOURCE address, the LDS image of a DMA is lane-linear -- so
//     that the 16 lanes of a ds_read_b128 group (16 different rows, same logical piece) hit 16 different 16-byte slots;
//   B slab: W' fragments in read order (prep16b), 1 contiguous KiB per (k-step, n-tile).
// Two slots: slab i + 1 travels while slab i is consumed (32 KB of dQKV16 in flight per CU: what keeps HBM streaming).
// The weight fragment is the MFMA's A operand and the data rows its B operand: the accumulator then holds four CONSECUTIVE
// output columns of one row per register quad, stored as 16 bytes (a quarter of the store instructions of the row-in-registers
// form).  (Round 2's kernel -- one wave per 32 rows, weights through a register-staged ring, A fragments straight from
// global -- ran at 0.37 ms: every tile step exposed a global-load latency.)
template <bool OUT16>
__global__ __launch_bounds__(DX_THREADS, 2) void gemm16_dx_kernel(Dx16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = a.m_dev != nullptr ? *a.m_dev : a.M;
    const int row0 = blockIdx.x * DX_BM;
    if (row0 >= M) return;                                              // whole workgroup beyond the rows (uniform)
    const float inv_scale = a.sc[1];
    // ---- this lane's share of the DMA fills.  A: piece p = wave + 8 i (i < 4) holds rows 8 p .. 8 p + 7; lane -> row 8 p + (lane >> 3),
    // physical 16-byte slot lane & 7, which receives the logical piece (lane & 7) ^ ((row >> 1) & 7).  Rows past the end are clamped
    // (computed, never stored).  B: piece q = wave + 8 i (i < 5), lane-linear.
    const char* asrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * (wave + 8 * i) + (lane >> 3);
        const int piece = (lane & 7) ^ ((r >> 1) & 7);
        asrc[i] = reinterpret_cast<const char*>(a.a16 + (long)min(row0 + r, M - 1) * a.lda) + piece * 16;
    }
    const char* bsrc = reinterpret_cast<const char*>(a.xtiles) + wave * 1024 + lane * 16;
    auto issue = [&](int slab, int slot) {
        char* la = smem + slot * DX_SLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + slab * (DX_BK * 2)),
                                             (__attribute__((address_space(3))) void*)(la + i * 8192), 16, 0, 0);
        char* lb = smem + slot * DX_SLOT + DX_A_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + (long)slab * DX_B_BYTES + i * 8192),
                                             (__attribute__((address_space(3))) void*)(lb + i * 8192), 16, 0, 0);
    };
    // fragment addresses inside a slot: rows wm * 64 + 32 i + l32 of A; tiles wn * 5 + j of B
    int aoff[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + 32 * i + l32;
#pragma unroll
        for (int s = 0; s < 4; ++s) aoff[i][s] = r * 128 + (((2 * s + hh) ^ ((r >> 1) & 7)) << 4);
    }
    const int boff = DX_A_BYTES + (wn * 5 * 64 + lane) * 16;            // + (s * 10 + j) * 1024

    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = zero16();
    issue(0, 0);
    __asm__ volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
    for (int slab = 0; slab < a.n_slabs; ++slab) {
        if (slab + 1 < a.n_slabs) issue(slab + 1, (slab + 1) & 1);
        const char* st = smem + (slab & 1) * DX_SLOT;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const h8 a0 = *reinterpret_cast<const h8*>(st + aoff[0][s]);
            const h8 a1 = *reinterpret_cast<const h8*>(st + aoff[1][s]);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const h8 b = *reinterpret_cast<const h8*>(st + boff + (s * 10 + j) * 1024);
                acc[0][j] = mfma32h(b, a0, acc[0][j]);               // D[n][row]: columns in registers, rows on lanes
                acc[1][j] = mfma32h(b, a1, acc[1][j]);
            }
        }
        // slab + 1 has landed (this wave's share: the counter; everybody's: the barrier) and nobody still reads this slot
        __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = row0 + wm * 64 + 32 * i + l32;
        if (row >= M) continue;
        float* crow = a.c + (long)row * a.ldc;
        _Float16* crow16 = reinterpret_cast<_Float16*>(a.c) + (long)row * F16_KP;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k = (wn * 5 + j) * 32 + 8 * g + 4 * hh;      // registers 4 g .. 4 g + 3 = columns k .. k + 3
                if (OUT16) {                                           // all 320 columns (those beyond d are exact zeros)
                    h4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (_Float16)acc[i][j][4 * g + e];
                    *reinterpret_cast<h4*>(crow16 + k) = v;
                } else if (k < a.d) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] * inv_scale;
                    *reinterpret_cast<f32x4*>(crow + k) = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C[n][k] = sum_m A[m][n] B[m][k] over the token rows m (fp16 operands, fp32 accumulate): d(W_qkv) = dQKV^T X and
// d(W_add) = dZ^T ctx.  The contraction runs over ROWS of both row-major operands, so each MFMA operand needs 8
// consecutive m of one column: 32 rows are staged in LDS as they arrive ([32 m][cols], pitch 8*odd dwords) and read back
// with ds_read_b64_tr_b16 (a 16-lane group reads 4 rows x 16 columns and gets them column-major).  8 waves = 4 (n) x 2 (k),
// NTN x NTK tiles of 16x16 per wave (v_mfma_f32_16x16x32_f16: one staged block = one k-step); M is split over
// workgroups into partial slabs, summed in fixed order by tn16_reduce_kernel (which also maps the internal row /
// column order back to the reference's parameter layout and removes the loss scale).
constexpr int T16_MC = 32, T16_THREADS = 512;
// Geometry of one instantiation: WN x WK = 8 waves, NTN x NTK tiles of 16 x 16 per wave; the workgroup's output block is
// AW x BW = (16 WN NTN) x (16 WK NTK).  d(W_qkv) [960 x 320]: 4 x 2 waves of 5 x 5 tiles = 320 x 160 blocks (3 x 2 of them);
// d(W_add) [224 x 320]: 2 x 4 waves of 7 x 5 tiles = ONE 224 x 320 block (with the 320 x 160 blocks 30 % of its MFMAs
// multiplied padding columns and every dZ row was staged twice).
template <int WN, int NTN, int WK, int NTK>
struct Tn16Geom {
    static_assert(WN * WK * 64 == T16_THREADS, "eight waves");
    static constexpr int AW = 16 * WN * NTN, BW = 16 * WK * NTK;       // staged columns per workgroup: n x k output block
    static constexpr int PA = AW + 16, PB = BW + 16;                   // pitches in halves: (AW + 16) / 2 dwords = 8 * odd
    static_assert(((PA / 2) % 16) == 8 && ((PB / 2) % 16) == 8, "pitch = 8 * odd dwords (conflict-free transposed reads)");
    static constexpr int A_BYTES = T16_MC * PA * 2, B_BYTES = T16_MC * PB * 2;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int A_IT = (T16_MC * AW / 8 + T16_THREADS - 1) / T16_THREADS;     // 16-byte chunks per thread
    static constexpr int B_IT = (T16_MC * BW / 8 + T16_THREADS - 1) / T16_THREADS;
};
typedef Tn16Geom<4, 5, 2, 5> Tn16Qkv;
typedef Tn16Geom<2, 7, 4, 5> Tn16Add;

struct Tn16Args {
    int M;                    // upper bound of the rows
    const int* m_dev;         // rows present (device) or null
    const _Float16* A; int lda, N;       // [M][lda], N used columns (multiple of 16)
    const _Float16* B; int ldb, K;       // [M][ldb], K used columns (multiple of 16)
    float* partial;           // [splits][N][K]
    int splits, n_blk, k_blk; // output blocks of AW x BW
    // fragment-order operands only: the 32-row blocks to contract over, as two of the title lists of launch_title_order
    // (blk_list[0 ..] and blk_list[2 blk_stride ..], sizes blk_cnt[0] and blk_cnt[2]: long and short titles -- the all-padding
    // titles' blocks are skipped); null: all M / 32 blocks in order
    const int* blk_list; const int* blk_cnt; int blk_stride;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ h8 tr16_frag(const char* plane, int pitch_b, int col0, int lane) {
    // operand fragment for columns col0..col0+15: element j of lane (c = lane&15, kq = lane>>4) = image[m(kq, j)][col0 + c],
    // m(kq, j) = 4 kq + j (j < 4), 16 + 4 kq + (j - 4)
    const int l16 = lane & 15, kq = lane >> 4;
    const int q = l16 >> 2, p = l16 & 3;
    const char* a0 = plane + (4 * kq + q) * pitch_b + (col0 + 4 * p) * 2;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 16 * pitch_b));
    s16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = lo4[i]; r[4 + i] = hi4[i]; }
    return __builtin_bit_cast(h8, r);
}

// FL: both operands are in the fragment order of fused16.h ([block of 32 rows][k-step][row][16]): a stage (32 rows) is ONE
// contiguous block per operand; otherwise plain row-major [M][ld].
template <bool FL, int WN, int NTN, int WK, int NTK>
__global__ __launch_bounds__(T16_THREADS, 2) void gemm16_tn_kernel(Tn16Args a) {
    typedef Tn16Geom<WN, NTN, WK, NTK> G;
    constexpr int T16_AW = G::AW, T16_BW = G::BW, T16_PA = G::PA, T16_PB = G::PB, T16_A_BYTES = G::A_BYTES;
    constexpr int T16_STAGE = G::STAGE, T16_A_IT = G::A_IT, T16_B_IT = G::B_IT, T16_NTN = NTN, T16_NTK = NTK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int out_blocks = a.n_blk * a.k_blk;
    // XCD-aware: workgroups are dealt round-robin over the 8 XCDs, and the out_blocks workgroups of one M-split read the
    // same rows (each A chunk k_blk times, each B chunk n_blk times): give an XCD a contiguous range of (split, block)
    // pairs so those re-reads hit its L2 (the grid is a multiple of 8)
    const int per_xcd = gridDim.x >> 3;
    const int idx = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    const int blk = idx % out_blocks, split = idx / out_blocks;
    const int bn = blk % a.n_blk, bk = blk / a.n_blk;
    const int ncol0 = bn * T16_AW, kcol0 = bk * T16_BW;
    const int n_list0 = (FL && a.blk_list != nullptr) ? a.blk_cnt[0] : 0;
    const int M = (FL && a.blk_list != nullptr) ? 32 * (n_list0 + a.blk_cnt[2]) : (a.m_dev != nullptr ? *a.m_dev : a.M);
    const int rps = (((M + a.splits - 1) / a.splits) + T16_MC - 1) / T16_MC * T16_MC;
    const int m_begin = split * rps, m_end = min(M, m_begin + rps);

    // staging slots (16-byte chunks); loads are unconditional from clamped addresses, zeroed when written to LDS.
    // A stage is only ~400 cycles of MFMA per wave, far less than one HBM round trip: the rows are prefetched TWO
    // stages ahead in two statically named register sets (the loop is unrolled by two; a run-time set index would go
    // through movrel), and the stage body is branch-free (clamped loads past the end, zeroed at the LDS write).
    struct Regs { h8 a[T16_A_IT], b[T16_B_IT]; };
    // slot -> (row r, column c) of the staged block.  Row-major source: a row's chunks are consecutive; fragment order:
    // consecutive slots walk (k-step, row, half) so that consecutive lanes read consecutive 16-byte pieces
    auto slot_rc = [&](int sl, int width, int& r, int& c) {
        if (FL) { r = (sl >> 1) & 31; c = (sl >> 6) * 16 + (sl & 1) * 8; if (c >= width) r = T16_MC; }
        else { r = sl / (width / 8); c = (sl % (width / 8)) * 8; }
    };
    auto load_stage = [&](Regs& R, int m0) {
        long blk = min(m0, max(m_end - 1, 0)) >> 5;                   // fragment order: the 32-row block of this stage
        if (FL && a.blk_list != nullptr && M > 0) blk = blk < n_list0 ? a.blk_list[blk] : a.blk_list[2 * (long)a.blk_stride + blk - n_list0];
#pragma unroll
        for (int i = 0; i < T16_A_IT; ++i) {
            int r, c;
            slot_rc(tid + T16_THREADS * i, T16_AW, r, c);
            r = min(r, T16_MC - 1);
            const int cc = min(ncol0 + c, a.N - 8);
            if (FL) R.a[i] = *reinterpret_cast<const h8*>(a.A + (blk * (a.lda >> 4) + (cc >> 4)) * 512 + r * 16 + (cc & 8));
            else R.a[i] = *reinterpret_cast<const h8*>(a.A + (long)min(m0 + r, max(m_end - 1, 0)) * a.lda + cc);
        }
#pragma unroll
        for (int i = 0; i < T16_B_IT; ++i) {
            int r, c;
            slot_rc(tid + T16_THREADS * i, T16_BW, r, c);
            r = min(r, T16_MC - 1);
            const int cc = min(kcol0 + c, a.K - 8);
            if (FL) R.b[i] = *reinterpret_cast<const h8*>(a.B + (blk * (a.ldb >> 4) + (cc >> 4)) * 512 + r * 16 + (cc & 8));
            else R.b[i] = *reinterpret_cast<const h8*>(a.B + (long)min(m0 + r, max(m_end - 1, 0)) * a.ldb + cc);
        }
    };
    const h8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto store_stage = [&](const Regs& R, int m0, char* st) {
#pragma unroll
        for (int i = 0; i < T16_A_IT; ++i) {
            int r, c;
            slot_rc(tid + T16_THREADS * i, T16_AW, r, c);
            if (r < T16_MC) {
                const bool ok = m0 + r < m_end && ncol0 + c < a.N;
                *reinterpret_cast<h8*>(st + (r * T16_PA + c) * 2) = ok ? R.a[i] : z8;
            }
        }
#pragma unroll
        for (int i = 0; i < T16_B_IT; ++i) {
            int r, c;
            slot_rc(tid + T16_THREADS * i, T16_BW, r, c);
            if (r < T16_MC) {
                const bool ok = m0 + r < m_end && kcol0 + c < a.K;
                *reinterpret_cast<h8*>(st + T16_A_BYTES + (r * T16_PB + c) * 2) = ok ? R.b[i] : z8;
            }
        }
    };
    f32x4 acc[T16_NTN][T16_NTK];
#pragma unroll
    for (int i = 0; i < T16_NTN; ++i)
#pragma unroll
        for (int j = 0; j < T16_NTK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const char* st) {
        h8 bf[T16_NTK];
#pragma unroll
        for (int j = 0; j < T16_NTK; ++j) bf[j] = tr16_frag(st + T16_A_BYTES, T16_PB * 2, (wk * T16_NTK + j) * 16, lane);
#pragma unroll
        for (int i = 0; i < T16_NTN; ++i) {
            const h8 af = tr16_frag(st, T16_PA * 2, (wn * T16_NTN + i) * 16, lane);
#pragma unroll
            for (int j = 0; j < T16_NTK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
        }
    };
    int n_stage = m_end > m_begin ? (m_end - m_begin + T16_MC - 1) / T16_MC : 0;
    if (n_stage > 0) {
        n_stage = (n_stage + 1) & ~1;                 // an odd count is rounded up (one all-zero stage)
        Regs R0, R1;
        load_stage(R1, m_begin);
        store_stage(R1, m_begin, smem);
        load_stage(R0, m_begin + T16_MC);
        __syncthreads();
        auto body = [&](const Regs& Rs, Regs& Rl, int s) {
            load_stage(Rl, m_begin + (s + 2) * T16_MC);
            store_stage(Rs, m_begin + (s + 1) * T16_MC, smem + ((s + 1) & 1) * T16_STAGE);
            compute(smem + (s & 1) * T16_STAGE);
            __syncthreads();
        };
        for (int s = 0; s < n_stage; s += 2) {
            body(R0, R1, s);
            body(R1, R0, s + 1);
        }
    }
    // partial slab [split][N][K]
    float* slab = a.partial + (long)split * a.N * a.K;
    const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < T16_NTN; ++i)
#pragma unroll
        for (int j = 0; j < T16_NTK; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int nn = ncol0 + (wn * T16_NTN + i) * 16 + 4 * kq + reg;
                const int kk = kcol0 + (wk * T16_NTK + j) * 16 + r16;
                if (nn < a.N && kk < a.K) slab[(long)nn * a.K + kk] = acc[i][j][reg];
            }
}

// dW[nmap[n]][kmap[k]] += scale[n] * sum_splits partial[.][n][k]   (maps: -1 = padding, dropped)
__global__ void tn16_reduce_kernel(const float* partial, int splits, int N, int K, const int* nmap, const int* kmap,
                                   const float* nscale, int ldw, float* dW, float* dbias) {
    const long total = (long)N * K;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int n = (int)(idx / K), k = (int)(idx - (long)n * K);
        const int nd = nmap[n], kd = kmap[k];
        if (nd < 0 || kd == -1 || (kd < 0 && dbias == nullptr)) continue;       // kd -2: the operand's ones column = the bias gradient
        const float* pp = partial + idx;
        // eight independent partial sums: eight loads in flight per thread (two made this kernel latency-bound); the
        // association is fixed, so the result is reproducible
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sp = 0;
        for (; sp + 8 <= splits; sp += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += pp[(long)(sp + u) * total];
        }
        for (; sp < splits; ++sp) s[0] += pp[(long)sp * total];
        const float tot = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) * nscale[n];
        if (kd >= 0) dW[(long)nd * ldw + kd] += tot;
        else dbias[nd] += tot;
    }
}

// bias / q_vec gradients from the per-workgroup column sums of the fused kernel (fixed order: 8 row groups per
// column, each summed in ascending workgroup order, then combined in a fixed tree)
__global__ __launch_bounds__(1024) void red16_kernel(const float* red, int n_wg, int h, int dk, int d, int q, const float* sc,
                                                    float qscale, float* db_qkv, float* db_add, float* dq_vec) {
    const float inv_scale = sc[1];
    __shared__ float part[32][33];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float s = 0.f;
    if (i < B16_RED) {
        // four loads in flight per thread (one made the kernel a chain of n_wg / 32 dependent loads); fixed association
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int g = rg;
        for (; g + 96 < n_wg; g += 128) {
            s0 += red[(long)g * B16_RED + i];
            s1 += red[(long)(g + 32) * B16_RED + i];
            s2 += red[(long)(g + 64) * B16_RED + i];
            s3 += red[(long)(g + 96) * B16_RED + i];
        }
        for (; g < n_wg; g += 32) s0 += red[(long)g * B16_RED + i];
        s = (s0 + s1) + (s2 + s3);
    }
    part[rg][c] = s;
    __syncthreads();
    if (rg != 0 || i >= B16_RED) return;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += part[r][c];               // fixed order
    s *= inv_scale;
    if (i < B16_RED_QKV) {
        const int tile = i >> 5, f = i & 31, head = tile / 3, which = tile - 3 * head;
        if (head < h && f < dk) db_qkv[which * d + head * dk + f] += s * (which == 0 ? qscale : 1.0f);
    } else if (i < B16_RED_QKV + F16_QP) {
        const int qq = i - B16_RED_QKV;
        if (qq < q) db_add[qq] += s;
    } else {
        const int qq = i - B16_RED_QKV - F16_QP;
        if (qq < q) dq_vec[qq] += s;
    }
}

// index maps of the reduce: rows of d(W_qkv) (dqkv16 column order), columns of d(W_add) (ctx16 column order)
__global__ void maps16_kernel(int d, int h, int dk, int q, const float* sc, float qscale, int* nmap_qkv, float* nscale_qkv,
                              int* kmap_x, int* nmap_add, float* nscale_add, int* kmap_ctx) {
    const float inv_scale = sc[1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B16_DQ) {
        const int head = i / 96, rem = i - head * 96, which = rem >> 5, p = rem & 31;
        const int s = p >> 4, t = p & 15, hh = t >> 3, jj = t & 7;
        const int f = 16 * s + 8 * (jj >> 2) + 4 * hh + (jj & 3);
        nmap_qkv[i] = (head < h && f < dk) ? which * d + head * dk + f : -1;
        nscale_qkv[i] = inv_scale * (which == 0 ? qscale : 1.0f);
    }
    if (i < F16_KP) kmap_x[i] = i < d ? i : -1;
    if (i < F16_QP) { nmap_add[i] = i < q ? i : -1; nscale_add[i] = inv_scale; }
    if (i < F16_DP) {
        const int b16 = i >> 4, t = i & 15, hh = t >> 3, jj = t & 7;
        const int fpad = 16 * b16 + 8 * (jj >> 2) + 4 * hh + (jj & 3);
        const int head = fpad >> 5, f = fpad & 31;
        kmap_ctx[i] = (head < h && f < dk) ? head * dk + f : -1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
static size_t up256(size_t x) { return (x + 255) / 256 * 256; }

// Two helper streams (a set per caller stream, side_streams_for in capi.hip) so that the two weight-gradient GEMMs --
// HBM-bound, independent of everything after them -- run beside the issue-bound attention kernel and the dX GEMM instead of
// behind them.  Forked from and joined back into the caller's stream with events inside the same C-ABI call: to the caller
// the call is still one in-order piece of work on its stream.  Events of the set: 0, 1 = forks, 2, 3 = joins (6, 7: guard).

Fused16BwdLayout fused16_bwd_layout(long M, int n_seq) {
    Fused16BwdLayout L;
    const long Mp = (long)n_seq * (M > (long)n_seq * 32 ? 64 : 32);      // fragment order: 32 (or 64) rows per sequence
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up256(off + bytes); return o; };
    L.n_wg = 512;
    L.btiles = take((size_t)40 * 32 * F16_KP * 2);
    L.xtiles = take((size_t)DX_SLABS * DX_B_BYTES);          // = 30 * 32 * KP * 2: the dX weight slabs
    L.qv16 = take((size_t)F16_QP * 2);
    L.bqkv32 = take((size_t)30 * 32 * 4);
    L.dout16 = take((size_t)n_seq * F16_DP * 2);
    L.dz16 = take((size_t)Mp * F16_QP * 2);
    L.dctx16 = take((size_t)Mp * F16_DP * 2);
    L.dqkv16 = take((size_t)(M + 32) * B16_DQ * 2);
    L.red = take((size_t)2 * L.n_wg * B16_RED * 4);      // pool kernel + attention kernel
    L.maps = take((size_t)(B16_DQ * 2 + F16_KP + F16_QP * 2 + F16_DP) * 4);
    L.scale = take(256);                                 // {scale, 1 / scale} floats, then max |dout| as uint bits
    // TN partial slabs.  The kernel's ~200 registers allow two waves per SIMD, i.e. exactly ONE 8-wave workgroup per CU:
    // the grid must not exceed the 256 CUs, or the surplus workgroups run as a second round on an otherwise idle GPU
    // (44 splits x 6 blocks = 264 workgroups took 0.39 ms alone, 40 x 6 = 240 take 0.23 ms; inside the step, where the
    // kernel shares the GPU with the main stream's kernels and they fill that idle round, the step time is the same), and
    // it stays a multiple of 8 for the XCD mapping (42 x 6 = 252 loses the L2 locality of the re-read rows: 0.29 ms)
    L.tn_splits_qkv = 40;      // x 6 output blocks of 320 x 160 = 240 workgroups
    // d(W_add) reads all of dZ16 and ctx16 once (1 GB: HBM-bound at ~4.5 TB/s whatever the tile shape -- the 224 x 320
    // block removed 30 % of its MFMAs and 40 % of its LDS traffic at unchanged time); 192 workgroups leave a quarter of the
    // CUs to the main stream's kernels it runs beside (same-box A/B: 128 / 192 / 256 splits within noise, 192 ahead)
    L.tn_splits_add = 192;     // x 1 output block of 224 x 320
    if (const char* e = getenv("NRMS_TN_SPLITS_QKV")) L.tn_splits_qkv = atoi(e);      // tuning only
    if (const char* e = getenv("NRMS_TN_SPLITS_ADD")) L.tn_splits_add = atoi(e);
    const size_t p1 = (size_t)L.tn_splits_qkv * B16_DQ * F16_KP * 4, p2 = (size_t)L.tn_splits_add * F16_QP * F16_DP * 4;
    L.partial = take(up256(p1) + p2);                    // both products have their own slabs: they run concurrently
    L.total = off;
    return L;
}

int launch_tn16(int geom, const _Float16* A, int lda, int N, const _Float16* B, int ldb, int K, int M, const int* m_dev,
                float* partial, int splits, const int* nmap, const int* kmap, const float* nscale, int ldw, float* dW,
                hipStream_t stream, const char* name, float* dbias, const int* blk_list, const int* blk_cnt, int blk_stride) {
    Tn16Args t{};
    t.M = M; t.m_dev = m_dev; t.A = A; t.lda = lda; t.N = N; t.B = B; t.ldb = ldb; t.K = K; t.partial = partial;
    t.blk_list = geom != 0 ? blk_list : nullptr; t.blk_cnt = blk_cnt; t.blk_stride = blk_stride;
    // geom 1 (fragment-order operands, one 224 x 320 block) = the additive product; 0 (row-major, 320 x 160 blocks) = the
    // Q|K|V product; 2 (fragment order, 320 x 160 blocks) = d(W_O) of nrms_v1
    typedef void (*Kern)(Tn16Args);
    const Kern fn = geom == 1 ? (Kern)gemm16_tn_kernel<true, 2, 7, 4, 5> : geom == 2 ? (Kern)gemm16_tn_kernel<true, 4, 5, 2, 5>
                                                                                     : (Kern)gemm16_tn_kernel<false, 4, 5, 2, 5>;
    const int aw = geom == 1 ? Tn16Add::AW : Tn16Qkv::AW, bw = geom == 1 ? Tn16Add::BW : Tn16Qkv::BW;
    t.splits = splits; t.n_blk = cdiv(N, aw); t.k_blk = cdiv(K, bw);
    const size_t lds = 2 * (size_t)(geom == 1 ? Tn16Add::STAGE : Tn16Qkv::STAGE);
    const hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return NRMS_ELAUNCH; }
    {
        TimingScope ts(name, stream);
        hipLaunchKernelGGL(fn, dim3(splits * t.n_blk * t.k_blk), dim3(T16_THREADS), lds, stream, t);
    }
    int rc = check_launch(name);
    if (rc) return rc;
    TimingScope ts("tn_reduce", stream);
    hipLaunchKernelGGL(tn16_reduce_kernel, dim3(cdiv((long)N * K, 256)), dim3(256), 0, stream, partial, splits, N, K, nmap, kmap,
                       nscale, ldw, dW, dbias);
    return check_launch("tn16_reduce");
}

size_t bwd16_fused_lds(bool two) {
    // ring + column sums + the per-wave staging rows of the atomic-free reductions (the pooling kernel's are the larger)
    static_assert(F16_WAVES * 2 * F16_QT * 32 + 2 * F16_WAVES * 32 >= 2 * F16_WAVES * 3 * 32, "the attention kernel's staging fits the same size");
    return (size_t)3 * F16_SLOT + (size_t)B16_RED * 4 + (size_t)(F16_WAVES * (two ? 2 : 1) * 2 * F16_QT * 32 + 2 * F16_WAVES * 32) * 4;
}

int launch_bwd16_pool(const Bwd16Args& a, int n_wg, bool two, hipStream_t stream) {
    const size_t lds = bwd16_fused_lds(two);
    const void* fp = two ? (const void*)fused_bwd16_pool_kernel<2> : (const void*)fused_bwd16_pool_kernel<1>;
    const hipError_t e = hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("fused_bwd16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    TimingScope ts(two ? "fused64_bwd16_pool" : "fused_bwd16_pool", stream);
    if (two) hipLaunchKernelGGL(fused_bwd16_pool_kernel<2>, dim3(n_wg), dim3(F16_THREADS), lds, stream, a);
    else hipLaunchKernelGGL(fused_bwd16_pool_kernel<1>, dim3(n_wg), dim3(F16_THREADS), lds, stream, a);
    return check_launch("fused_bwd16_pool");
}

int launch_dx16(const Dx16Args& g, bool out16, hipStream_t stream) {
    if ((g.d & 3) != 0 || ((uintptr_t)g.c & 15) != 0) { set_error("gemm16_dx: dx must be 16-byte aligned, d %% 4 == 0"); return NRMS_EINVAL; }
    const size_t lds = (size_t)2 * DX_SLOT;
    const void* fn = out16 ? (const void*)gemm16_dx_kernel<true> : (const void*)gemm16_dx_kernel<false>;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("gemm16_dx: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    TimingScope ts("dx_bwd", stream);
    if (out16) hipLaunchKernelGGL(gemm16_dx_kernel<true>, dim3(cdiv(g.M, DX_BM)), dim3(DX_THREADS), lds, stream, g);
    else hipLaunchKernelGGL(gemm16_dx_kernel<false>, dim3(cdiv(g.M, DX_BM)), dim3(DX_THREADS), lds, stream, g);
    return check_launch("gemm16_dx");
}

int launch_dout16(long n_seq, int d, int h, int dk, float fixed_scale, float* sc, const float* dout, _Float16* dout16, hipStream_t stream) {
    unsigned* max_bits = (unsigned*)(sc + 2);
    // loss scale: from max |dout| on the device unless the caller fixed it (fixed_scale > 0)
    if (hipMemsetAsync(max_bits, 0, sizeof(unsigned), stream) != hipSuccess) { set_error("fused_bwd16: memset failed"); return NRMS_ELAUNCH; }
    const long nd = n_seq * d;
    if (!(fixed_scale > 0.f))
        hipLaunchKernelGGL(absmax_kernel, dim3(nd > 512L * 1024 ? 512 : (int)cdiv(nd, 1024)), dim3(256), 0, stream, nd, dout, max_bits);
    hipLaunchKernelGGL(dout16_kernel, dim3(cdiv(n_seq * F16_DP, 256) > 4096 ? 4096 : cdiv(n_seq * F16_DP, 256)),
                       dim3(256), 0, stream, n_seq, d, h, dk, fixed_scale, max_bits, sc, dout, dout16);
    return check_launch("dout16");
}

int launch_fused_bwd16(const Fused16Bwd& f, hipStream_t stream) {
    if (f.n_seq <= 0) return NRMS_OK;
    const long M = (long)f.n_seq * f.S;
    const Fused16BwdLayout L = fused16_bwd_layout(M, f.n_seq);
    char* base = (char*)f.workspace;
    const int dk = f.d / f.h;
    const float qscale = 1.0f / sqrtf((float)dk);
    _Float16* btiles = (_Float16*)(base + L.btiles);
    _Float16* xtiles = (_Float16*)(base + L.xtiles);
    _Float16* qv16 = (_Float16*)(base + L.qv16);
    float* bqkv32 = (float*)(base + L.bqkv32);
    _Float16* dout16 = (_Float16*)(base + L.dout16);
    _Float16* dz16 = (_Float16*)(base + L.dz16);
    _Float16* dctx16 = (_Float16*)(base + L.dctx16);
    _Float16* dqkv16 = (_Float16*)(base + L.dqkv16);
    float* red = (float*)(base + L.red);
    int* nmap_qkv = (int*)(base + L.maps);
    float* nscale_qkv = (float*)(nmap_qkv + B16_DQ);
    int* kmap_x = (int*)(nscale_qkv + B16_DQ);
    int* nmap_add = kmap_x + F16_KP;
    float* nscale_add = (float*)(nmap_add + F16_QP);
    int* kmap_ctx = (int*)(nscale_add + F16_QP);
    float* sc = (float*)(base + L.scale);                      // device: {scale, 1 / scale}
    if (f.sc_out != nullptr) *f.sc_out = sc;
    float* partial_qkv = (float*)(base + L.partial);
    float* partial_add = (float*)(base + L.partial + up256((size_t)L.tn_splits_qkv * B16_DQ * F16_KP * 4));
    { const int jr = fused_bwd16_join(stream); if (jr) return jr; }      // an un-joined earlier call on this stream: order it first
    // (NRMS_NO_SIDE_STREAMS is read per call so that a profiler pass can serialise the step: exclusive kernel durations)
    SideSet* ss = side_streams_for(stream);
    const bool side = ss != nullptr;
    hipStream_t s_add = side ? ss->s[0] : stream, s_qkv = side ? ss->s[1] : stream;
    SideJoinGuard join_guard(ss, stream);
    const int Mp = f.n_seq * 32 * (f.S > 32 ? 2 : 1);         // fragment order: 32 rows per block, zero beyond a sequence
    {
        Prep16bArgs p{};
        p.d = f.d; p.h = f.h; p.dk = dk; p.q = f.q; p.w_qkv = f.w_qkv; p.b_qkv = f.b_qkv; p.w_add = f.w_add; p.q_vec = f.q_vec;
        p.btiles = btiles; p.xtiles = xtiles; p.qv16 = qv16; p.bqkv32 = bqkv32;
        TimingScope ts("prep16", stream);
        hipLaunchKernelGGL(prep16b_kernel, dim3(2048), dim3(256), 0, stream, p);
        int rc0 = launch_dout16((long)f.n_seq, f.d, f.h, dk, f.loss_scale, sc, f.dout, dout16, stream);
        if (rc0) return rc0;
        hipLaunchKernelGGL(maps16_kernel, dim3(cdiv(B16_DQ, 256)), dim3(256), 0, stream, f.d, f.h, dk, f.q, sc, qscale,
                           nmap_qkv, nscale_qkv, kmap_x, nmap_add, nscale_add, kmap_ctx);
        int rc = check_launch("prep16b");
        if (rc) return rc;
    }
    Bwd16Args a{};
    a.n_seq = f.n_seq; a.S = f.S; a.d = f.d; a.h = f.h; a.dk = dk; a.q = f.q;
    a.n_groups = cdiv(f.n_seq, F16_WAVES) + (f.order != nullptr ? 2 : 0);      // three lists: up to two more partial groups
    a.n_cls = f.order != nullptr ? 3 : 2;                                       // (launch_title_order(..., 3) by the caller)
    a.x16 = (const _Float16*)f.x16; a.pos = f.pos; a.n_rows = f.n_rows_dev; a.ids = f.ids; a.order = f.order; a.order_cnt = f.order_cnt;
    a.btiles = btiles; a.bqkv32 = bqkv32; a.qv16 = qv16;
    a.ctx16 = (const _Float16*)f.ctx16; a.t16 = (const _Float16*)f.t16; a.w = f.w; a.dout16 = dout16;
    a.dz16 = dz16; a.dctx16 = dctx16; a.dqkv16 = dqkv16; a.red = red; a.drop = f.drop;
    a.dbg = 0;
#ifdef NRMS_F16_EXPERIMENTS
    { const char* e = getenv("NRMS_F16_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
    const int n_wg = a.n_groups < L.n_wg ? a.n_groups : L.n_wg;
    {
        const bool two = f.S > 32;
        const size_t lds = bwd16_fused_lds(two);
        const void* fa = two ? (const void*)fused_bwd16_attn_kernel<2> : (const void*)fused_bwd16_attn_kernel<1>;
        const hipError_t e = hipFuncSetAttribute(fa, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("fused_bwd16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
        {
            const int rcp = launch_bwd16_pool(a, n_wg, two, stream);
            if (rcp) return rcp;
        }
        if (side) {
            // d(W_add)[q][f] = sum_tok dZ[tok][q] ctx[tok][f] needs only the pooling kernel's dZ16: beside the attention kernel
            int rc = side_order(ss, 0, stream, s_add, "fused_bwd16");
            if (rc) return rc;
            rc = launch_tn16(1, dz16, F16_QP, F16_QP, (const _Float16*)f.ctx16, F16_DP, F16_DP, Mp, nullptr, partial_add,
                             L.tn_splits_add, nmap_add, kmap_ctx, nscale_add, f.d, f.dw_add, s_add, "dwadd_bwd");
            if (rc) return rc;
            if (hipEventRecord(ss->ev[2], s_add) != hipSuccess) { set_error("fused_bwd16: hipEventRecord failed"); return NRMS_ELAUNCH; }
        }
        Bwd16Args b = a;
        b.red = red + (long)n_wg * B16_RED;                       // second set of per-workgroup sums
        {
            TimingScope ts(two ? "fused64_bwd16_attn" : "fused_bwd16_attn", stream);
            if (two) hipLaunchKernelGGL(fused_bwd16_attn_kernel<2>, dim3(n_wg), dim3(F16_THREADS), lds, stream, b);
            else hipLaunchKernelGGL(fused_bwd16_attn_kernel<1>, dim3(n_wg), dim3(F16_THREADS), lds, stream, b);
        }
        int rc = check_launch("fused_bwd16");
        if (rc) return rc;
    }
    {
        TimingScope ts("red16", stream);
        hipLaunchKernelGGL(red16_kernel, dim3(cdiv(B16_RED, 32)), dim3(1024), 0, stream, red, 2 * n_wg, f.h, dk, f.d, f.q, sc,
                           qscale, f.db_qkv, f.db_add, f.dq_vec);
        int rc = check_launch("red16");
        if (rc) return rc;
    }
    int rc = NRMS_OK;
    if (!side) {
        rc = launch_tn16(1, dz16, F16_QP, F16_QP, (const _Float16*)f.ctx16, F16_DP, F16_DP, Mp, nullptr, partial_add,
                         L.tn_splits_add, nmap_add, kmap_ctx, nscale_add, f.d, f.dw_add, stream, "dwadd_bwd");
        if (rc) return rc;
    }
    // d(W_qkv)[n][k] = sum_rows dQKV[r][n] x[r][k]   (live rows only): beside the dX GEMM
    if (side) {
        rc = side_order(ss, 1, stream, s_qkv, "fused_bwd16");
        if (rc) return rc;
    }
    rc = launch_tn16(0, dqkv16, B16_DQ, B16_DQ, (const _Float16*)f.x16, F16_KP, F16_KP, (int)M, f.n_rows_dev, partial_qkv,
                     L.tn_splits_qkv, nmap_qkv, kmap_x, nscale_qkv, f.d, f.dw_qkv, s_qkv, "dwqkv_bwd");
    if (rc) return rc;
    if (side && hipEventRecord(ss->ev[3], s_qkv) != hipSuccess) { set_error("fused_bwd16: hipEventRecord failed"); return NRMS_ELAUNCH; }
    // dX = dQKV W'  -> fp32 rows
    {
        Dx16Args g{};
        g.M = (int)M; g.m_dev = f.n_rows_dev; g.a16 = dqkv16; g.xtiles = xtiles; g.c = f.dx; g.ldc = f.d; g.d = f.d;
        g.sc = sc; g.lda = B16_DQ; g.n_slabs = DX_SLABS;
        rc = launch_dx16(g, f.dx_fp16, stream);
    }
    if (side && rc == NRMS_OK) {                                    // join: the caller's stream continues after both GEMMs
        if (f.defer_join) ss->pending = true;                       // ... or later, in fused_bwd16_join
        else if (hipStreamWaitEvent(stream, ss->ev[2], 0) != hipSuccess || hipStreamWaitEvent(stream, ss->ev[3], 0) != hipSuccess) {
            set_error("fused_bwd16: hipStreamWaitEvent failed");
            return NRMS_ELAUNCH;                                    // (the guard still joins what it can)
        }
        join_guard.disarm();
    }
    return rc;
}

// The other half of Fused16Bwd::defer_join: work enqueued on `stream` after this sees d(W_qkv), d(b_qkv), d(W_add), d(b_add).
int fused_bwd16_join(hipStream_t stream) {
    SideSet* ss = side_streams_for(stream);
    if (ss == nullptr || !ss->pending) return NRMS_OK;
    ss->pending = false;
    if (hipStreamWaitEvent(stream, ss->ev[2], 0) != hipSuccess || hipStreamWaitEvent(stream, ss->ev[3], 0) != hipSuccess) {
        set_error("fused_bwd16_join: hipStreamWaitEvent failed");
        return NRMS_ELAUNCH;
    }
    return NRMS_OK;
}

}  // namespace nrms
