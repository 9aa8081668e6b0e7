// fp16 mode (NRMS_PRECISION_FP16): the encoder forward as ONE kernel, one wavefront per sequence.
//
//   ids -> [gather16: x16 fp16, compact live rows] -> fused_fwd16:
//       per head:  QT = Wq x^T, KT = Wk x^T (features x tokens), V = x Wv^T (tokens x features)   -- v_mfma_f32_32x32x16_f16
//                  S^T = K Q^T  -> softmax over the key ROWS (registers)  -> ctx^T = V^T P^T      -- operands straight
//                  from the accumulators of the previous product (no LDS, no shuffles)
//       then:      T^T = Wadd ctx^T, tanh, . q_vec (register rows), softmax over the tokens (lanes), pooling
//
// Replaces NewsEncoder.forward / UserEncoder.forward (model/nrms_v0.py:154-176, 188-199; SDPA :13-23, MHSA :46-76,
// additive attention :100-126) for sequences of at most 32 rows.  The only things shared between the 8 waves of a
// workgroup are the weight tiles (32 output rows x K, fp16), streamed through a 3-slot LDS ring; Q, K, V, the
// attention probabilities and tanh(.) never leave the register file.  HBM sees: x16 (read), ctx16 / T16 / w (written
// for the backward), the [n_seq, d] output.
//
// Layouts (all fp16 tensors are zero padded):
//   x16    [rows][KP]      KP = 320 (fixed pitch, d <= 320); row = compact live-token index (pos[token]) or the token itself
//   (each tile in the LDS-DMA order of fused16.h: [k-step][lane half][row][8], dma_tile_pos)
//   wqkv16 [3h tiles][32][KP]   tile 3*head + {0,1,2} = the head's W_Q (pre-scaled by 1/sqrt(d_k)), W_K, W_V rows
//   ctx16  [n_seq*S][DP]   DP = 320 >= 32 h: head-padded features, and INSIDE every 16-feature block in "P16" order: memory
//                          position 8*hh + j  <->  feature 16 b + 8 (j>>2) + 4 hh + (j&3).  That is exactly the order in
//                          which a 32x32 accumulator hands its rows to the next MFMA as an operand (cdna guide, section 3),
//                          so ctx^T goes from the PV product into the additive projection without any data movement.
//   wadd16 [QP/32 tiles][32][DP]  columns in the same P16 order (QP = 224 >= q);   T16 [n_seq*S][QP] natural order
#include <stdlib.h>

#include "fused16.h"

namespace nrms {

struct Fwd16Args {
    int n_seq, S, d, h, dk, q;
    const _Float16* x16;      // [rows][KP]
    const int* pos;           // [n_seq*S] token -> x16 row, -1 = padding token (-> the pad row); null: row = token
    const int* n_rows;        // with pos: number of compact rows (device); row *n_rows of x16 is the padding token's row
                              // (zeros and the ones column: its Q|K|V is exactly the bias)
    const int64_t* ids;       // news encoder with NRMS_FLAG_PAD_ROW_ZERO: all-padding titles take the closed form; else null
    const int* order;         // optional [2][n_seq]: the titles with a real token, then (second row) the all-padding
    const int* order_cnt;     //          titles; order_cnt[0..1] = their numbers (device).  null = identity
    const _Float16* wtiles;   // [3h + QP/32][32][KP]: the head tiles (Q pre-scaled | K | V per head), then the additive tiles
    const float* bqkv32;      // [3h][32]  (Q part pre-scaled)
    const float* badd32;      // [QP]
    const float* qv32;        // [QP]
    _Float16* ctx16;          // [n_seq*S][DP]
    _Float16* t16;            // [n_seq*S][QP] or null (inference)
    float* w;                 // [n_seq*S] or null
    float* out;               // [n_seq][d]
    Dropout drop;             // context dropout (site 1, element index = token * DP + padded feature)
    int dbg;                  // timing experiments only (NRMS_F16_DBG): 1 = no tile staging after the prologue
};

// SB = 32-row blocks per sequence: 1 (titles, histories of at most 32 slots; two workgroups per CU) or 2 (sequences of up
// to 64 rows, e.g. the 50-slot histories of the user encoder: one wave still owns the whole sequence, with twice the
// accumulators -- one workgroup per CU, up to 512 registers per lane; it is 3 % of the flops, what matters is that its
// activations stay on chip like the titles')
#ifndef F16_FWD_SLOTS
#define F16_FWD_SLOTS 3          // tuning: 3 slots / 2 workgroups per CU; 2 slots / 3 workgroups per CU (168 VGPRs, 16 spilled) measured the same
#endif
constexpr int F16_FWD_WGS = F16_FWD_SLOTS == 2 ? 3 : 2;
template <bool TRAIN, int SB>
__global__ __launch_bounds__(F16_THREADS, SB == 1 ? F16_FWD_WGS : 1) void fused_fwd16_kernel(Fwd16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int S = a.S;
    constexpr int KP = F16_KP, DP = F16_DP;
    const int n_head_tiles = 3 * a.h;

    // ---- which sequence this wave owns.  With an order list, workgroups [0, ceil(n_ne / 4)) take the titles that
    // have a real token, the following ones the all-padding titles (closed form: no head tile is touched), so no
    // wave idles through the head tiles next to a working one.
    int slot_id = blockIdx.x * F16_WAVES + wave;
    bool valid = slot_id < a.n_seq;
    int seq = slot_id;
    if (a.order != nullptr) {
        const int n_ne = a.order_cnt[0], n_e = a.order_cnt[1];
        const int g_ne = (n_ne + F16_WAVES - 1) / F16_WAVES;
        if ((int)blockIdx.x < g_ne) { valid = slot_id < n_ne; seq = valid ? a.order[slot_id] : 0; }
        else {
            slot_id -= g_ne * F16_WAVES;
            valid = slot_id < n_e;
            seq = valid ? a.order[a.n_seq + slot_id] : 0;
        }
    }
    if (!valid) seq = 0;
    const long tok0 = (long)seq * S;                             // first token of the sequence
    bool tok_ok[SB];
#pragma unroll
    for (int b = 0; b < SB; ++b) tok_ok[b] = valid && 32 * b + l32 < S;
    bool empty = false;                                          // all-padding title: attention is uniform over equal rows
    if (SB == 1 && a.ids != nullptr && valid) {
        const bool is_pad = lane < S ? a.ids[tok0 + lane] == 0 : true;
        empty = __ballot(is_pad) == ~0ull;
    }
    const bool live = valid && !empty;
    const bool skip_heads = __syncthreads_and(live ? 0 : 1) != 0;     // whole workgroup without a live title
    const int n_begin = skip_heads ? n_head_tiles : 0;

    using Ring = TileRingDMA<SB == 1 ? F16_FWD_SLOTS : 3>;
    constexpr int AH = Ring::AHEAD;
    Ring ring;
    ring.smem = smem; ring.src = a.wtiles; ring.n_tiles = n_head_tiles + F16_QT; ring.wave = wave; ring.lane = lane; ring.l32 = l32; ring.hh = hh;
#pragma unroll
    for (int i = 0; i < AH; ++i) ring.load(n_begin + i);
    // the additive stage's per-row vectors in LDS behind the ring (see fused_fwd16p_kernel; visible after the prologue's barrier)
    float* addv = reinterpret_cast<float*>(smem + (SB == 1 ? F16_FWD_SLOTS : 3) * F16_SLOT_DMA);      // [2][F16_QP]
    for (int i = tid; i < 2 * F16_QP; i += F16_THREADS) addv[i] = i < F16_QP ? a.badd32[i] : a.qv32[i - F16_QP];

    // ---- this lane's x fragments: token 32 b + l32, features 16 s + 8 hh .. +7 (A operand of x W^T, B operand of W x^T)
    h8 xf[SB][F16_KS];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        long row = -1;
        if (tok_ok[b] && !empty) {
            row = a.pos != nullptr ? (long)a.pos[tok0 + 32 * b + l32] : tok0 + 32 * b + l32;
            if (row < 0) row = *a.n_rows;                       // padding token inside a live title
        }
        const _Float16* xr = a.x16 + (row < 0 ? 0 : row) * KP + 8 * hh;
#pragma unroll
        for (int s = 0; s < F16_KS; ++s) xf[b][s] = *reinterpret_cast<const h8*>(xr + 16 * s);
        if (row < 0) {
#pragma unroll
            for (int s = 0; s < F16_KS; ++s) xf[b][s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // the first two tiles and the x fragments have landed.  The wait is the BUILTIN (vmcnt(0), encoded for gfx9): hipcc's
    // wait-count pass does not see inline assembly, would still believe the x-fragment loads pending at the loop header and
    // put a vmcnt(0) in front of every tile's first MFMA -- draining the DMA of the tile after next on every step
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int n = n_begin;
#pragma unroll 1
    for (int head = 0; head < a.h; ++head) {
        f32x16 ct[SB];                                           // ctx^T[f][tok] of this head, per token block
        if (!skip_heads) {
            f32x16 qt[SB], kt[SB], vv[SB];
            // ---- tile Q: QT[f][tok] = sum_k Wq[f][k] x[tok][k] (+ b through the ones column)
            auto pre = [&](int g) { ring.load_group(n + AH, g); };    // a later tile travels while tile n is consumed
            if (!live) ring.load(n + AH);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                qt[b] = zero16();
                if (live) { if (b == 0) tile_mma<true>(qt[b], ring, n, xf[b], pre); else tile_mma<true>(qt[b], ring, n, xf[b]); }
            }
            ring.step_barrier(n);
            ++n;
            // ---- tile K
            if (!live) ring.load(n + AH);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                kt[b] = zero16();
                if (live) { if (b == 0) tile_mma<true>(kt[b], ring, n, xf[b], pre); else tile_mma<true>(kt[b], ring, n, xf[b]); }
            }
            ring.step_barrier(n);
            ++n;
            // ---- tile V: V[tok][f] = sum_k x[tok][k] Wv[f][k]
            if (!live) ring.load(n + AH);
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                vv[b] = zero16();
                if (live) { if (b == 0) tile_mma<false>(vv[b], ring, n, xf[b], pre); else tile_mma<false>(vv[b], ring, n, xf[b]); }
            }
            // ---- attention of this head, entirely in registers: one query block at a time
            if (live) {
#pragma unroll
                for (int ib = 0; ib < SB; ++ib) {
                    f32x16 st[SB];                               // S^T[j][i]: rows j = keys (block jb) in registers, columns i = queries
                    float m = -3.0e38f;
#pragma unroll
                    for (int jb = 0; jb < SB; ++jb) {
                        st[jb] = mfma32h(acc_frag(kt[jb], 0), acc_frag(qt[ib], 0), zero16());
                        st[jb] = mfma32h(acc_frag(kt[jb], 1), acc_frag(qt[ib], 1), st[jb]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            st[jb][r] = 32 * jb + crow32(r, hh) < S ? st[jb][r] : -3.0e38f;   // rows beyond the sequence are not keys
                            m = fmaxf(m, st[jb][r]);
                        }
                    }
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    float sum = 0.f;
#pragma unroll
                    for (int jb = 0; jb < SB; ++jb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float p = 32 * jb + crow32(r, hh) < S ? __expf(st[jb][r] - m) : 0.f;
                            st[jb][r] = p;
                            sum += p;
                        }
                    sum += __shfl_xor(sum, 32, 64);
                    const float inv = 1.0f / sum;
                    // ctx^T[f][i] = sum_j V[j][f] P^T[j][i]
                    ct[ib] = zero16();
#pragma unroll
                    for (int jb = 0; jb < SB; ++jb) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) st[jb][r] *= inv;
                        ct[ib] = mfma32h(acc_frag(vv[jb], 0), acc_frag(st[jb], 0), ct[ib]);
                        ct[ib] = mfma32h(acc_frag(vv[jb], 1), acc_frag(st[jb], 1), ct[ib]);
                    }
                }
            }
        }
        if (!live) {
#pragma unroll
            for (int b = 0; b < SB; ++b) ct[b] = rows_of(a.bqkv32 + (3 * head + 2) * 32, hh);   // all-padding title: ctx = b_v
        }
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            if (a.drop.thresh != 0u) {
                // two Philox calls for the lane's 16 values: registers 8c..8c+7 take the 8 fields of group
                // (token row, 16-column half c of the head, lane half) -- see nrms_dropout_keep_mask, site 1 | 0x100
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const uint64_t e0 = (uint64_t)(tok0 + 32 * b + l32) * (uint64_t)DP + (uint64_t)(head * 32 + 16 * c + 8 * hh);
                    float sc[8];
                    dropout_scale8(a.drop.seed, 1u, e0 >> 3, a.drop.thresh16, a.drop.inv_keep, sc);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ct[b][8 * c + e] *= sc[e];
                }
            }
            if (valid && !F16_DBG(a.dbg, 8)) {                        // rows beyond the sequence are stored too: zeros
                const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                _Float16* dst = a.ctx16 + frag_off((long)seq * SB + b, F16_CS, 2 * head, l32, hh);
                *reinterpret_cast<h8*>(dst) = tok_ok[b] ? acc_frag(ct[b], 0) : z;
                *reinterpret_cast<h8*>(dst + 512) = tok_ok[b] ? acc_frag(ct[b], 1) : z;
            }
        }
        if (!skip_heads) {
            ring.step_barrier(n);          // (counting this head's 2 SB context stores as "younger" measured no gain: kept at the safe 0)
            ++n;
        }
    }
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        if (valid) {                                               // heads the model does not have: zero columns
            for (int head = a.h; head < F16_CS / 2; ++head) {
                _Float16* dst = a.ctx16 + frag_off((long)seq * SB + b, F16_CS, 2 * head, l32, hh);
                *reinterpret_cast<h8*>(dst) = h8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<h8*>(dst + 512) = h8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }

    // ---- additive attention: T^T[q][tok] = sum_f Wadd[q][f] ctx^T[f][tok]; the ctx operand is read back in the
    // very layout it was stored in (own stores: wait for them, then plain loads)
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's ctx16 stores have landed
    h8 cf[SB][F16_CS];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        const _Float16* src = a.ctx16 + frag_off((long)seq * SB + b, F16_CS, 0, l32, hh);      // (zeros beyond the sequence)
#pragma unroll
        for (int s = 0; s < F16_CS; ++s) cf[b][s] = *reinterpret_cast<const h8*>(src + (F16_DBG(a.dbg, 16) ? 0 : 512 * s));
    }
    float score[SB];                                              // sum_q q_vec[q] tanh(.)[q][tok], per token = per lane
#pragma unroll
    for (int b = 0; b < SB; ++b) score[b] = 0.f;
#pragma unroll 1
    for (int t = 0; t < F16_QT; ++t) {                            // (not unrolled: hipcc would software-pipeline the tanh
                                                                  //  epilogues across tiles and spill their accumulators)
        // tanh(y) = 1 - 2 / (exp(2 y) + 1) on v_exp / v_rcp: ~1e-7 absolute, far inside what fp16 keeps of it;
        // badd32 holds b * 2 log2(e), so exp(2 (x + b)) = exp2(x * c + b')
        const f32x16 ba = rows_of(addv + 32 * t, hh), qq = rows_of(addv + F16_QP + 32 * t, hh);
        __builtin_amdgcn_sched_barrier(0);
        auto pre2 = [&](int g) { ring.load_group(n + AH, g); };
        if (!valid) ring.load(n + AH);
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            f32x16 tt = zero16();
            if (valid) { if (b == 0) tile_mma<true>(tt, ring, n, cf[b], pre2); else tile_mma<true>(tt, ring, n, cf[b]); }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h4 th;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ex = __builtin_amdgcn_exp2f(fmaf(tt[4 * g + e], 2.885390082f, ba[4 * g + e]));
                    const float v = fmaf(-2.0f, __builtin_amdgcn_rcpf(ex + 1.0f), 1.0f);
                    score[b] += qq[4 * g + e] * v;
                    th[e] = (_Float16)v;
                }
                if (TRAIN && valid && !F16_DBG(a.dbg, 4)) {
                    if (!tok_ok[b]) th = h4{0, 0, 0, 0};
                    *reinterpret_cast<h4*>(a.t16 + ((((long)seq * SB + b) * (F16_QP / 16) + 2 * t + (g >> 1)) * 32 + l32) * 16 +
                                           8 * (g & 1) + 4 * hh) = th;
                }
            }
        }
        ring.step_barrier(n);
        ++n;
    }
    // softmax over the tokens of the sequence (lane l32 of block b holds token 32 b + l32, in either half)
    float mx = -3.0e38f;
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        score[b] += __shfl_xor(score[b], 32, 64);
        score[b] = 32 * b + l32 < S ? score[b] : -3.0e38f;
        mx = fmaxf(mx, score[b]);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float es = 0.f, wgt[SB];
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        wgt[b] = 32 * b + l32 < S ? __expf(score[b] - mx) : 0.f;
        es += wgt[b];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) es += __shfl_xor(es, o, 64);
#pragma unroll
    for (int b = 0; b < SB; ++b) {
        wgt[b] /= es;
        if (TRAIN && a.w != nullptr && tok_ok[b] && hh == 0) a.w[tok0 + 32 * b + l32] = wgt[b];
    }

    // ---- pooling: out[f] = sum_tok w_tok ctx[tok][f], from the ctx fragments already in registers.  A product with a
    // column selector moves the tokens from the lanes into the accumulator's rows (D[tok][n] = ctx[tok][32 p + n]),
    // where the weighted sum over tokens is a sum over registers (+ one lane-half exchange) in fp32.
    if (valid) {
        h8 sel[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) sel[s2][j] = (_Float16)(l32 == 16 * s2 + 8 * hh + j ? 1.0f : 0.0f);
        float wrow[SB][16];                                      // w of the token held in register r of this lane half
#pragma unroll
        for (int b = 0; b < SB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) wrow[b][r] = __shfl(wgt[b], crow32(r, hh), 64);
#pragma unroll
        for (int p = 0; p < F16_CS / 2; ++p) {                   // 32 padded features = one head per step (unrolled: cf is indexed)
            float acc = 0.f;
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                // cf element j of half hh is padded feature 16 s + 8 (j>>2) + 4 hh + (j&3) (P16 order): as the A operand its k
                // index 8 hh + j is routed by the selector to column n = 16 s2 + 8 hh + j, i.e. D[tok][n] = ctx[tok][position n]
                f32x16 dd = mfma32h(cf[b][2 * p], sel[0], zero16());
                dd = mfma32h(cf[b][2 * p + 1], sel[1], dd);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc += wrow[b][r] * dd[r];
            }
            acc += __shfl_xor(acc, 32, 64);
            // column n = l32 holds P16 position n of head p: padded feature 16 (n>>4) + 8 ((n&7)>>2) + 4 ((n>>3)&1) + (n&3)
            const int n = l32, f = 16 * (n >> 4) + 8 * ((n & 7) >> 2) + 4 * ((n >> 3) & 1) + (n & 3);
            if (hh == 0 && f < a.dk && p < a.h) a.out[(long)seq * a.d + p * a.dk + f] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The news encoder's forward on the padding-skipping path (NRMS_FLAG_PAD_ROW_ZERO: pos / ids / a 3-class order list):
// the same products as fused_fwd16_kernel<., 1>, with the 32-row tile of the ATTENTION stage used twice as well.
//
// A padding token's Q | K | V row is the bias, the same for every padding token of a title, so in the attention stage the
// L - n padding tokens of a title with n real tokens collapse exactly: as keys into ONE key whose logit gains log(L - n),
// as queries into one row whose context they all share.  A "short" title (its n <= 15 real tokens a prefix) therefore needs
// n + 1 <= 16 rows, and TWO short titles share one tile: rows 0..15 title A, rows 16..31 title B, with the cross-title entries
// of S^T masked.  Per pair the 30 head tiles cost what they cost one title before (66 MFMAs per head, one softmax); a MIND
// title is ~11 words, the bench's are 5..19: three quarters of the titles pair up.  The additive stage is per token -- the
// context dropout draws a different mask for each of the L tokens, padding or not -- so each title expands back to its L rows
// (lane t of title p pulls column 16 p + min(t, n_p) of the context with ds_bpermute), goes through the dropout, the ctx16
// store and the seven additive tiles exactly as before; a pair walks the additive tiles twice.
//   workgroups [0, g_pair): 8 short titles (2 per wave) | [g_pair, +g_long): 4 long titles | then 4 all-padding titles
// The softmax mask is an additive per-register bias (0 / log multiplicity / -3e38) prepared once per title: one add per
// element instead of the compare + select pair of the general kernel.
template <bool TRAIN>
__global__ __launch_bounds__(F16_THREADS, F16_FWD_WGS) void fused_fwd16p_kernel(Fwd16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int S = a.S;
    constexpr int KP = F16_KP, DP = F16_DP;
    const int n_head_tiles = 3 * a.h;
    constexpr float NEG = -3.0e38f;

    // ---- which titles this wave owns
    const int n_long = a.order_cnt[0], n_e = a.order_cnt[1], n_short = a.order_cnt[2];
    const int g_pair = (n_short + 2 * F16_WAVES - 1) / (2 * F16_WAVES);
    const int g_long = (n_long + F16_WAVES - 1) / F16_WAVES, g_e = (n_e + F16_WAVES - 1) / F16_WAVES;
    const int blk = blockIdx.x;
    if (blk >= g_pair + g_long + g_e) return;                       // surplus workgroup (the grid is an upper bound)
    const bool pair = blk < g_pair, skip_heads = blk >= g_pair + g_long;      // uniform over the workgroup
    const int NT = pair ? 2 : 1;                                    // titles per wave
    int seq[2] = {0, 0};
    bool val[2] = {false, false};
    if (pair) {
        const int p0 = 2 * (blk * F16_WAVES + wave);
#pragma unroll
        for (int i = 0; i < 2; ++i) { val[i] = p0 + i < n_short; seq[i] = val[i] ? a.order[2 * (long)a.n_seq + p0 + i] : 0; }
    } else if (!skip_heads) {
        const int sl = (blk - g_pair) * F16_WAVES + wave;
        val[0] = sl < n_long; seq[0] = val[0] ? a.order[sl] : 0;
    } else {
        const int sl = (blk - g_pair - g_long) * F16_WAVES + wave;
        val[0] = sl < n_e; seq[0] = val[0] ? a.order[(long)a.n_seq + sl] : 0;
    }
    long tok0[2];
    int nlive[2];                                                   // real tokens of a short title (a prefix by classification)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        tok0[i] = (long)seq[i] * S;
        nlive[i] = 0;
        if (pair && val[i]) nlive[i] = __popcll(__ballot(lane < S && a.ids[tok0[i] + lane] != 0));
    }

    // ---- the tile's rows: (title, token) of row l32, its x16 row, and the softmax bias of the key rows this lane holds
    const int myp = pair ? (l32 >> 4) : 0, myr = pair ? (l32 & 15) : l32;
    const int myn = myp ? nlive[1] : nlive[0];
    const bool myval = myp ? val[1] : val[0];
    long xrow = -1;                                                 // -1: a row of zeros (rows the tile does not use)
    if (!skip_heads && myval) {
        const long t0 = myp ? tok0[1] : tok0[0];
        if (pair) {
            if (myr < myn) xrow = a.pos[t0 + myr];                  // a real token (pos >= 0 by classification)
            else if (myr == myn && myn < S) xrow = *a.n_rows;       // the title's padding tokens, all in this one row
        } else if (myr < S) {
            xrow = a.pos[t0 + myr];
            if (xrow < 0) xrow = *a.n_rows;                         // a padding token inside a long title: x16's padding row
        }
    }
    f32x16 kbias;                                                   // per key row j = crow32(r, hh), for this lane's query column l32
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = crow32(r, hh);
        float bsv;
        if (pair) {
            const int jr = j & 15;
            const float lm = __logf((float)max(S - myn, 1));
            bsv = (j >> 4) != myp ? NEG : (jr < myn ? 0.f : ((jr == myn && myn < S) ? lm : NEG));
        } else {
            bsv = j < S ? 0.f : NEG;
        }
        kbias[r] = bsv;
    }
    // expansion: token l32 of title i reads the context column of its row (all padding tokens: the shared row)
    int src_lane[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) src_lane[i] = 4 * (16 * i + min(l32, nlive[i]) + 32 * hh);
    bool tok_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) tok_ok[i] = val[i] && l32 < S;

    // ---- tile steps: the 3 h head tiles (not for all-padding titles), then the 7 additive tiles once per title
    const int n_head_steps = skip_heads ? 0 : n_head_tiles;
    const int n_steps = n_head_steps + F16_QT * NT;
    auto tile_at = [&](int step) { return step < n_head_steps ? step : n_head_tiles + (step - n_head_steps) % F16_QT; };
    using Ring = TileRingDMA<F16_FWD_SLOTS>;
    constexpr int AH = Ring::AHEAD;
    Ring ring;
    ring.smem = smem; ring.src = a.wtiles; ring.n_tiles = n_steps; ring.wave = wave; ring.lane = lane; ring.l32 = l32; ring.hh = hh;
#pragma unroll
    for (int i = 0; i < AH; ++i) ring.load_at(i, tile_at(i));       // (n_steps >= 7 > AH)
    // the additive stage's per-row vectors (bias, query vector) live in LDS behind the ring: a global load per tile step in the
    // additive loop puts a vmcnt wait -- which also drains the step's own stores -- in front of every tile's epilogue
    float* addv = reinterpret_cast<float*>(smem + F16_FWD_SLOTS * F16_SLOT_DMA);          // [2][F16_QP]
    for (int i = tid; i < 2 * F16_QP; i += F16_THREADS) addv[i] = i < F16_QP ? a.badd32[i] : a.qv32[i - F16_QP];

    h8 xf[F16_KS];
    {
        const _Float16* xr = a.x16 + (xrow < 0 ? 0 : xrow) * KP + 8 * hh;
#pragma unroll
        for (int s = 0; s < F16_KS; ++s) xf[s] = *reinterpret_cast<const h8*>(xr + 16 * s);
        if (xrow < 0) {
#pragma unroll
            for (int s = 0; s < F16_KS; ++s) xf[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): see fused_fwd16_kernel
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const bool any_live = val[0] || val[1];                         // this wave has work in the head tiles
    int n = 0;
#pragma unroll 1
    for (int head = 0; head < a.h; ++head) {
        f32x16 ct;                                                  // ctx^T[f][tile column] of this head
        if (!skip_heads) {
            f32x16 qt = zero16(), kt = zero16(), vv = zero16();
            auto pre = [&](int g) { if (n + AH < n_steps) ring.load_piece_at(n + AH, tile_at(n + AH), g); };
            if (!any_live && n + AH < n_steps) ring.load_at(n + AH, tile_at(n + AH));
            if (any_live) tile_mma<true>(qt, ring, n, xf, pre);
            ring.step_barrier(n);
            ++n;
            if (!any_live && n + AH < n_steps) ring.load_at(n + AH, tile_at(n + AH));
            if (any_live) tile_mma<true>(kt, ring, n, xf, pre);
            ring.step_barrier(n);
            ++n;
            if (!any_live && n + AH < n_steps) ring.load_at(n + AH, tile_at(n + AH));
            if (any_live) tile_mma<false>(vv, ring, n, xf, pre);
            if (any_live) {
                // S^T[j][i] (rows j = keys in registers, columns i = queries), masked / weighted by the additive bias
                f32x16 st = mfma32h(acc_frag(kt, 0), acc_frag(qt, 0), zero16());
                st = mfma32h(acc_frag(kt, 1), acc_frag(qt, 1), st);
                float m = NEG;
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[r] += kbias[r]; m = fmaxf(m, st[r]); }
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] - m); sum += st[r]; }
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] *= inv;
                ct = mfma32h(acc_frag(vv, 0), acc_frag(st, 0), zero16());
                ct = mfma32h(acc_frag(vv, 1), acc_frag(st, 1), ct);
            } else {
                ct = zero16();
            }
        } else {
            ct = rows_of(a.bqkv32 + (3 * head + 2) * 32, hh);       // all-padding title: ctx = b_v
        }
        // ---- per title: its L rows of the context, dropout, ctx16
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < NT) {
                f32x16 cx = ct;
                if (pair) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        cx[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane[i], __float_as_int(ct[r])));
                }
                if (a.drop.thresh != 0u) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const uint64_t e0 = (uint64_t)(tok0[i] + l32) * (uint64_t)DP + (uint64_t)(head * 32 + 16 * c + 8 * hh);
                        float sc[8];
                        dropout_scale8(a.drop.seed, 1u, e0 >> 3, a.drop.thresh16, a.drop.inv_keep, sc);
#pragma unroll
                        for (int e = 0; e < 8; ++e) cx[8 * c + e] *= sc[e];
                    }
                }
                if (val[i]) {                                       // rows beyond the sequence are stored too: zeros
                    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    _Float16* dst = a.ctx16 + frag_off((long)seq[i], F16_CS, 2 * head, l32, hh);
                    *reinterpret_cast<h8*>(dst) = tok_ok[i] ? acc_frag(cx, 0) : z;
                    *reinterpret_cast<h8*>(dst + 512) = tok_ok[i] ? acc_frag(cx, 1) : z;
                }
            }
        }
        if (!skip_heads) {
            ring.step_barrier(n);
            ++n;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i < NT && val[i]) {                                     // heads the model does not have: zero columns
            for (int head = a.h; head < F16_CS / 2; ++head) {
                _Float16* dst = a.ctx16 + frag_off((long)seq[i], F16_CS, 2 * head, l32, hh);
                *reinterpret_cast<h8*>(dst) = h8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<h8*>(dst + 512) = h8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }

    // ---- additive attention + pooling, one title after the other (see fused_fwd16_kernel for the commentary)
#pragma unroll 1
    for (int i = 0; i < NT; ++i) {
        const bool valid = i ? val[1] : val[0];
        const bool tok = i ? tok_ok[1] : tok_ok[0];
        const int sq = i ? seq[1] : seq[0];
        const long t0 = i ? tok0[1] : tok0[0];
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's ctx16 stores have landed
        h8 cf[F16_CS];
        {
            const _Float16* src = a.ctx16 + frag_off((long)sq, F16_CS, 0, l32, hh);
#pragma unroll
            for (int s = 0; s < F16_CS; ++s) cf[s] = *reinterpret_cast<const h8*>(src + 512 * s);
        }
        float score = 0.f;
#pragma unroll 1
        for (int t = 0; t < F16_QT; ++t) {
            const f32x16 ba = rows_of(addv + 32 * t, hh), qq = rows_of(addv + F16_QP + 32 * t, hh);
            __builtin_amdgcn_sched_barrier(0);
            auto pre2 = [&](int g) { if (n + AH < n_steps) ring.load_piece_at(n + AH, tile_at(n + AH), g); };
            if (!valid && n + AH < n_steps) ring.load_at(n + AH, tile_at(n + AH));
            f32x16 tt = zero16();
            if (valid) tile_mma<true>(tt, ring, n, cf, pre2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                h4 th;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ex = __builtin_amdgcn_exp2f(fmaf(tt[4 * g + e], 2.885390082f, ba[4 * g + e]));
                    const float v = fmaf(-2.0f, __builtin_amdgcn_rcpf(ex + 1.0f), 1.0f);
                    score += qq[4 * g + e] * v;
                    th[e] = (_Float16)v;
                }
                if (TRAIN && valid) {
                    if (!tok) th = h4{0, 0, 0, 0};
                    *reinterpret_cast<h4*>(a.t16 + (((long)sq * (F16_QP / 16) + 2 * t + (g >> 1)) * 32 + l32) * 16 + 8 * (g & 1) + 4 * hh) = th;
                }
            }
            ring.step_barrier(n);
            ++n;
        }
        score += __shfl_xor(score, 32, 64);
        score = l32 < S ? score : NEG;
        float mx = score;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float wgt = l32 < S ? __expf(score - mx) : 0.f;
        float es = wgt;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) es += __shfl_xor(es, o, 64);
        wgt /= es;
        if (TRAIN && a.w != nullptr && tok && hh == 0) a.w[t0 + l32] = wgt;
        if (valid) {
            h8 sel[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) sel[s2][j] = (_Float16)(l32 == 16 * s2 + 8 * hh + j ? 1.0f : 0.0f);
            float wrow[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) wrow[r] = __shfl(wgt, crow32(r, hh), 64);
#pragma unroll
            for (int p = 0; p < F16_CS / 2; ++p) {
                f32x16 dd = mfma32h(cf[2 * p], sel[0], zero16());
                dd = mfma32h(cf[2 * p + 1], sel[1], dd);
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc += wrow[r] * dd[r];
                acc += __shfl_xor(acc, 32, 64);
                const int nn = l32, f = 16 * (nn >> 4) + 8 * ((nn & 7) >> 2) + 4 * ((nn >> 3) & 1) + (nn & 3);
                if (hh == 0 && f < a.dk && p < a.h) a.out[(long)sq * a.d + p * a.dk + f] = acc;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight planes of one encoder (a few hundred KB): one launch per call
struct Prep16Args {
    int d, h, dk, q, KP, DP, QP;
    const float* w_qkv;   // [3d][d]
    const float* b_qkv;   // [3d]
    const float* w_add;   // [q][d]
    const float* b_add;   // [q]
    const float* q_vec;   // [q]
    _Float16* wqkv16;     // [3h][32][KP]
    float* bqkv32;        // [3h][32]
    _Float16* wadd16;     // [QP][DP]  (P16 column order)
    float* badd32;        // [QP]
    float* qv32;          // [QP]
};

__global__ __launch_bounds__(256) void prep16_kernel(Prep16Args a) {
    const float qscale = 1.0f / sqrtf((float)a.dk);
    const long n1 = (long)3 * a.h * 32 * a.KP, n2 = (long)a.QP * a.DP, n3 = (long)3 * a.h * 32, n4 = a.QP;
    const long total = n1 + n2 + n3 + 2 * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (i < n1) {
            const int k = (int)(i % a.KP);
            const long r = i / a.KP;
            const int f = (int)(r & 31), tile = (int)(r >> 5), head = tile / 3, which = tile - 3 * head;
            float v = 0.f;
            if (f < a.dk && k < a.d) v = a.w_qkv[((long)which * a.d + head * a.dk + f) * a.d + k] * (which == 0 ? qscale : 1.0f);
            if (f < a.dk && k == a.d) v = a.b_qkv[which * a.d + head * a.dk + f] * (which == 0 ? qscale : 1.0f);   // x16[:, d] = 1
            a.wqkv16[(long)tile * 32 * a.KP + dma_tile_pos(f, k)] = (_Float16)v;
        } else if (i < n1 + n2) {
            const long j = i - n1;
            const int p = (int)(j % a.DP), qq = (int)(j / a.DP);
            // memory position p of the row holds padded feature fpad (P16 order inside each 16-block)
            const int b16 = p >> 4, t = p & 15, hh = t >> 3, jj = t & 7;
            const int fpad = 16 * b16 + 8 * (jj >> 2) + 4 * hh + (jj & 3);
            const int head = fpad >> 5, f = fpad & 31;
            float v = 0.f;
            if (qq < a.q && f < a.dk) v = a.w_add[(long)qq * a.d + head * a.dk + f];
            a.wadd16[(long)(qq >> 5) * 32 * a.DP + dma_tile_pos(qq & 31, p)] = (_Float16)v;
        } else if (i < n1 + n2 + n3) {
            const long j = i - n1 - n2;
            const int f = (int)(j & 31), tile = (int)(j >> 5), head = tile / 3, which = tile - 3 * head;
            float v = 0.f;
            if (f < a.dk) v = a.b_qkv[which * a.d + head * a.dk + f] * (which == 0 ? qscale : 1.0f);
            a.bqkv32[j] = v;
        } else if (i < n1 + n2 + n3 + n4) {
            const long j = i - n1 - n2 - n3;
            a.badd32[j] = j < a.q ? a.b_add[j] * 2.885390082f : 0.f;          // pre-multiplied by 2 log2(e) (see the tanh)
        } else {
            const long j = i - n1 - n2 - n3 - n4;
            a.qv32[j] = j < a.q ? a.q_vec[j] : 0.f;
        }
    }
}

// Title lists, each IN ASCENDING TITLE ORDER (a stable partition: which workgroup sums which titles' bias gradients -- and
// therefore the last bits of those sums -- must not depend on how a race between workgroups resolves):
//   NCLS = 2:  order[0][..] = titles with at least one non-padding token, order[1][..] = all-padding titles;
//   NCLS = 3:  order[0][..] = "long" titles, order[1][..] = all-padding titles, order[2][..] = "short" titles: the non-padding
//              tokens are a PREFIX of 1 .. 15 tokens, so that with one row for all its padding tokens the title fits 16 rows
//              and two of them share one 32-row tile (fused_fwd16p_kernel).
// cnt[0 .. NCLS) = the list sizes, cnt[NCLS + NCLS b + c] = the count of class c in block b (256 titles).
// 256 titles per workgroup (64 per wave): the id loads of 16 titles are issued together (one dependent load per title made
// the kernel a 16-long latency chain).  Pass 1 (PLACE = false) writes the per-block counts, pass 2 recomputes the flags (the
// ids are 6.7 MB: cheaper than a flag array round trip), sums the counts of the blocks before it and places its titles.
template <int NCLS, bool PLACE>
__global__ __launch_bounds__(256) void title_order_kernel(int n_seq, int S, const int64_t* ids, int* order, int* cnt) {
    __shared__ int flag[256];
    __shared__ int wcount[4][NCLS], base[NCLS], part[4][NCLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = blockIdx.x * 256;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        int64_t v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {           // unconditional loads from clamped addresses: a predicated load is a branch + wait each
            const int t = min(t0 + wave * 64 + r * 16 + i, n_seq - 1);
            v[i] = ids[(long)t * S + min(lane, S - 1)];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = t0 + wave * 64 + r * 16 + i;
            const unsigned long long livemask = __ballot(lane < S && v[i] != 0);
            int cls = livemask == 0ull ? 1 : 0;
            if (NCLS == 3 && livemask != 0ull) {
                const int n = __popcll(livemask);
                if (n <= 15 && livemask == (1ull << n) - 1ull) cls = 2;
            }
            if (lane == 0) flag[wave * 64 + r * 16 + i] = t < n_seq ? cls : -1;
        }
    }
    __syncthreads();
    const int f = flag[threadIdx.x];
    const unsigned long long below = (1ull << lane) - 1ull;
    int rank = 0;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        const unsigned long long m = __ballot(f == c);
        if (f == c) rank = __popcll(m & below);
        if (lane == 0) wcount[wave][c] = __popcll(m);
    }
    if (PLACE) {
        // counts of the blocks before this one (and, in the last block, of all blocks: the totals)
        int s[NCLS];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) s[c] = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256)
#pragma unroll
            for (int c = 0; c < NCLS; ++c) s[c] += cnt[NCLS + NCLS * b + c];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s[c] += __shfl_xor(s[c], o, 64);
            if (lane == 0) part[wave][c] = s[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < NCLS) {
        const int c = threadIdx.x;
        const int mine = wcount[0][c] + wcount[1][c] + wcount[2][c] + wcount[3][c];
        if (!PLACE) cnt[NCLS + NCLS * blockIdx.x + c] = mine;
        else {
            base[c] = part[0][c] + part[1][c] + part[2][c] + part[3][c];
            if (blockIdx.x == gridDim.x - 1) cnt[c] = base[c] + mine;
        }
    }
    if (!PLACE) return;
    __syncthreads();
    if (f >= 0) {
        for (int w = 0; w < wave; ++w) rank += wcount[w][f];
        order[(long)f * n_seq + base[f] + rank] = t0 + threadIdx.x;
    }
}

size_t title_order_cnt_ints(int n_seq) { return 3 + 3 * (size_t)cdiv(n_seq > 0 ? n_seq : 1, 256); }

// n_classes = 2 (order [2][n_seq]) or 3 (order [3][n_seq])
int launch_title_order(int n_seq, int S, const int64_t* ids, int* order, int* cnt, hipStream_t stream, int n_classes) {
    if (n_seq <= 0) return NRMS_OK;
    TimingScope ts("title_order", stream);
    const dim3 grid(cdiv(n_seq, 256));
    if (n_classes == 3) {
        hipLaunchKernelGGL((title_order_kernel<3, false>), grid, dim3(256), 0, stream, n_seq, S, ids, order, cnt);
        hipLaunchKernelGGL((title_order_kernel<3, true>), grid, dim3(256), 0, stream, n_seq, S, ids, order, cnt);
    } else {
        hipLaunchKernelGGL((title_order_kernel<2, false>), grid, dim3(256), 0, stream, n_seq, S, ids, order, cnt);
        hipLaunchKernelGGL((title_order_kernel<2, true>), grid, dim3(256), 0, stream, n_seq, S, ids, order, cnt);
    }
    return check_launch("title_order");
}

// x16[r, :] = fp16(table[ids[t], :] * keep(t, :) / (1 - p)), t = live[r] (r < *n_live) or t = r (live == null);
// columns d..KP-1 are zero.  Same Philox counters (site 0, element index t*d + c) as the fp32 gather.
__global__ __launch_bounds__(256) void gather16_kernel(unsigned d4, unsigned kp4, long M, const int64_t* ids, const int* live,
                                                       const int* n_live, const float* table, Dropout drop, _Float16* x16) {
    const unsigned rows = live != nullptr ? (unsigned)(*n_live) : (unsigned)M;
    const unsigned total = (rows + (live != nullptr ? 1u : 0u)) * kp4;         // compact: + the padding token's row
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const unsigned r = i / kp4, c4 = i - r * kp4;
        h4 o = {0, 0, 0, 0};
        if (c4 == d4) o[0] = (_Float16)1.0f;                                    // the ones column (bias of Q|K|V)
        if (r >= rows) { *reinterpret_cast<h4*>(x16 + (long)i * 4) = o; continue; }
        const long t = live != nullptr ? (long)live[r] : (long)r;
        if (c4 < d4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(table + ids[t] * (long)(4 * d4) + 4 * c4);
            if (drop.thresh != 0u) v *= dropout_scale4(drop.seed, 0u, (uint64_t)(t * d4 + c4), drop.thresh, drop.inv_keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
        }
        *reinterpret_cast<h4*>(x16 + (long)i * 4) = o;
    }
}

// fp32 rows [M][d] (the user encoder's input: news vectors) -> x16 [M][KP]
__global__ __launch_bounds__(256) void cast16_kernel(unsigned d4, unsigned kp4, long M, const float* x, _Float16* x16) {
    const unsigned total = (unsigned)M * kp4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / kp4, c4 = i - r * kp4;
        h4 o = {0, 0, 0, 0};
        if (c4 == d4) o[0] = (_Float16)1.0f;                                    // the ones column
        if (c4 < d4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((long)r * d4 + c4) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
        }
        *reinterpret_cast<h4*>(x16 + (long)i * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------
bool fused16_supported(int S, int d, int h, int q, const char** why) {
    const int dk = d / h;
    const char* w = nullptr;
    if (S > 64) w = "seq_len <= 64";
    else if (d > F16_KP - 4) w = "d_model <= 316";          // one spare column carries the bias of Q|K|V
    else if (dk > 32) w = "d_k <= 32";
    else if (32 * h > F16_DP) w = "n_heads <= 10";
    else if (q > F16_QP) w = "q_dim <= 224";
    if (why) *why = w;
    return w == nullptr;
}

Fused16Layout fused16_layout(int d, int h, int q) {
    Fused16Layout L;
    L.KP = F16_KP;                 // fixed pitches: see the constants at the top
    L.DP = F16_DP;
    L.QP = F16_QP;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    L.wqkv16 = take((size_t)3 * h * 32 * L.KP * 2 + (size_t)L.QP * L.DP * 2);     // head tiles, then the additive tiles: ONE tile stream
    L.wadd16 = L.wqkv16 + (size_t)3 * h * 32 * L.KP * 2;
    L.bqkv32 = take((size_t)3 * h * 32 * 4);
    L.badd32 = take((size_t)L.QP * 4);
    L.qv32 = take((size_t)L.QP * 4);
    L.total = off;
    return L;
}

int launch_prep16(int d, int h, int q, const float* w_qkv, const float* b_qkv, const float* w_add, const float* b_add,
                  const float* q_vec, void* planes, hipStream_t stream) {
    const Fused16Layout L = fused16_layout(d, h, q);
    char* base = (char*)planes;
    Prep16Args a{};
    a.d = d; a.h = h; a.dk = d / h; a.q = q; a.KP = L.KP; a.DP = L.DP; a.QP = L.QP;
    a.w_qkv = w_qkv; a.b_qkv = b_qkv; a.w_add = w_add; a.b_add = b_add; a.q_vec = q_vec;
    a.wqkv16 = (_Float16*)(base + L.wqkv16); a.wadd16 = (_Float16*)(base + L.wadd16);
    a.bqkv32 = (float*)(base + L.bqkv32); a.badd32 = (float*)(base + L.badd32); a.qv32 = (float*)(base + L.qv32);
    const long total = (long)3 * h * 32 * L.KP + (long)L.QP * L.DP + 3 * h * 32 + 2 * L.QP;
    TimingScope ts("prep16", stream);
    hipLaunchKernelGGL(prep16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);
    return check_launch("prep16");
}

int launch_gather16(long M, int d, int KP, const int64_t* ids, const int* live, const int* n_live, const float* table,
                    const Dropout& drop, void* x16, hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    if ((M + 1) * (KP / 4) >= (1L << 32)) { set_error("gather16: index overflow"); return NRMS_EINVAL; }
    int blocks = cdiv((M + 1) * (KP / 4), 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("gather_dropout", stream);
    hipLaunchKernelGGL(gather16_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned)(d / 4), (unsigned)(KP / 4), M, ids, live,
                       n_live, table, drop, (_Float16*)x16);
    return check_launch("gather16");
}

int launch_cast16(long M, int d, int KP, const float* x, void* x16, hipStream_t stream) {
    if (M <= 0) return NRMS_OK;
    if (M * (KP / 4) >= (1L << 32)) { set_error("cast16: index overflow"); return NRMS_EINVAL; }
    int blocks = cdiv(M * (KP / 4), 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    TimingScope ts("cast16", stream);
    hipLaunchKernelGGL(cast16_kernel, dim3(blocks), dim3(256), 0, stream, (unsigned)(d / 4), (unsigned)(KP / 4), M, x, (_Float16*)x16);
    return check_launch("cast16");
}

int launch_fused_fwd16(const Fused16Fwd& f, hipStream_t stream) {
    if (f.n_seq <= 0) return NRMS_OK;
    const Fused16Layout L = fused16_layout(f.d, f.h, f.q);
    const char* base = (const char*)f.planes;
    Fwd16Args a{};
    a.n_seq = f.n_seq; a.S = f.S; a.d = f.d; a.h = f.h; a.dk = f.d / f.h; a.q = f.q;
    a.x16 = (const _Float16*)f.x16; a.pos = f.pos; a.n_rows = f.n_rows; a.ids = f.ids; a.order = f.order; a.order_cnt = f.order_cnt;
    a.wtiles = (const _Float16*)(base + L.wqkv16); a.bqkv32 = (const float*)(base + L.bqkv32);
    a.badd32 = (const float*)(base + L.badd32); a.qv32 = (const float*)(base + L.qv32);
    a.ctx16 = (_Float16*)f.ctx16; a.t16 = (_Float16*)f.t16; a.w = f.w; a.out = f.out; a.drop = f.drop;
    const bool train = f.t16 != nullptr, two = f.S > 32;
    const size_t lds = (size_t)(two ? 3 : F16_FWD_SLOTS) * F16_SLOT_DMA + (size_t)2 * F16_QP * 4;     // (+ the paired kernel's additive vectors)
    a.dbg = 0;
#ifdef NRMS_F16_EXPERIMENTS
    { const char* e = getenv("NRMS_F16_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
    // padding-skipping path with a 3-class order list: the kernel that pairs short titles (NRMS_NO_PAIRING: A/B switch, read per call)
    const bool paired = !two && f.order != nullptr && f.pos != nullptr && f.ids != nullptr;
    if (!paired) { a.order = nullptr; a.order_cnt = nullptr; }      // (the general kernel takes titles in index order)
    const void* fn = paired ? (train ? (const void*)fused_fwd16p_kernel<true> : (const void*)fused_fwd16p_kernel<false>)
                   : two ? (train ? (const void*)fused_fwd16_kernel<true, 2> : (const void*)fused_fwd16_kernel<false, 2>)
                         : (train ? (const void*)fused_fwd16_kernel<true, 1> : (const void*)fused_fwd16_kernel<false, 1>);
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("fused_fwd16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    TimingScope ts(two ? "fused64_fwd16" : "fused_fwd16", stream);
    const dim3 grid(cdiv(f.n_seq, F16_WAVES) + (f.order != nullptr ? 3 : 0));    // three lists: up to three partial groups
    if (paired) {
        if (train) hipLaunchKernelGGL((fused_fwd16p_kernel<true>), grid, dim3(F16_THREADS), lds, stream, a);
        else hipLaunchKernelGGL((fused_fwd16p_kernel<false>), grid, dim3(F16_THREADS), lds, stream, a);
    } else if (two) {
        if (train) hipLaunchKernelGGL((fused_fwd16_kernel<true, 2>), grid, dim3(F16_THREADS), lds, stream, a);
        else hipLaunchKernelGGL((fused_fwd16_kernel<false, 2>), grid, dim3(F16_THREADS), lds, stream, a);
    } else {
        if (train) hipLaunchKernelGGL((fused_fwd16_kernel<true, 1>), grid, dim3(F16_THREADS), lds, stream, a);
        else hipLaunchKernelGGL((fused_fwd16_kernel<false, 1>), grid, dim3(F16_THREADS), lds, stream, a);
    }
    return check_launch("fused_fwd16");
}

}  // namespace nrms
