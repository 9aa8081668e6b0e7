// fp16 mode for the nrms_v1 news encoder (model/nrms_v1.py:109-162): multi-head self-attention with heads wider than 32
// (title_heads_num = 6 -> d_k = 50) and the output projection W_O, dropout AFTER W_O (nrms_v1.py:161), additive attention.
// Same one-wavefront-per-title design as fused16.hip (two short titles per tile, 3-class title lists), with every weight tile
// still a [32 x 320] tile of the shared ring:
//
//   * a head is two 32-column feature blocks (features 0..31 | 32..d_k-1, zero padded): tiles Q0 K0 Q1 K1 V0 V1 per head,
//     S^T = K0 Q0^T + K1 Q1^T (4 MFMAs), one softmax, ctx^T per block = V_b^T P^T;
//   * the head concatenation ("attn", d = h d_k = 300 wide) is stored in 20 k-steps of 16: three full k-steps per head
//     (features 0..47), ONE k-step collecting the heads' remaining features (48, 49 of each of the six heads), and the
//     k-step of the ones column (the bias of W_O rides in it, as the bias of Q|K|V rides in x16's) -- so that
//   * W_O is ten ordinary tiles over that 320-wide operand, its output rows grouped 30 per tile: the result IS v0's ctx16
//     layout (ten "virtual heads" of 30), the dropout is v0's context dropout, and the additive stage, the pooling, the
//     pooling backward and d(W_add) of fused16.hip / fused16_bwd.hip run unchanged on it;
//   * an all-padding title: every attn row is b_v, so its pre-dropout row is the constant W_O b_v + b_o (prepared once per
//     call), and it walks the additive tiles only.
//
// Covered: the padding-skipping path (NRMS_FLAG_PAD_ROW_ZERO), seq_len <= 32, d <= 316 with d / hv <= 32 for hv = 10 output
// blocks, 32 < d_k <= 50 with h (d_k - 48) <= 16 leftover features and 3 h + 1 <= 19 k-steps -- the reference's v1
// configuration (config.py:87-88: six title heads of 50 on 300-wide embeddings).  Anything else stays on the bf16x3 kernels.
#include "fused16_v1.h"

namespace nrms {

bool fused16v1_supported(int S, int d, int h, int q, const char** why) {
    const char* w = nullptr;
    const int dk = h > 0 ? d / h : 0;
    if (S > 32) w = "seq_len <= 32";
    else if (d > F16_KP - 4 || d % 10 != 0 || d / 10 > 32) w = "d_model <= 316, a multiple of 10 with d_model / 10 <= 32";
    else if (d % h != 0 || dk <= 32 || dk > 50) w = "32 < d_k <= 50";
    else if (3 * h + 1 > 19 || h * (dk > 48 ? dk - 48 : 0) > 16) w = "3 h + 1 <= 19 k-steps and h (d_k - 48) <= 16 leftover features";
    else if (q > F16_QP) w = "q_dim <= 224";
    if (why) *why = w;
    return w == nullptr;
}

// ---- planes: [6 h head tiles | hv W_O tiles | 7 additive tiles] (each 32 x 320 fp16 in DMA order), boeff32, badd32, qv32
struct Fused16V1Layout { size_t tiles, boeff32, badd32, qv32, total; int n_tiles; };
Fused16V1Layout fused16v1_layout(int d, int h, int q) {
    const V1Geom g = v1_geom(d, h);
    Fused16V1Layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    L.n_tiles = g.n_head_tiles + g.hv + F16_QT;
    L.tiles = take((size_t)L.n_tiles * 32 * F16_KP * 2);
    L.boeff32 = take((size_t)g.hv * 32 * 4);
    L.badd32 = take((size_t)F16_QP * 4);
    L.qv32 = take((size_t)F16_QP * 4);
    L.total = off;
    return L;
}

struct Prep16V1Args {
    V1Geom g;
    int q;
    const float *w_qkv, *b_qkv, *w_o, *b_o, *w_add, *b_add, *q_vec;
    _Float16* tiles;
    float *boeff32, *badd32, *qv32;
};

__global__ __launch_bounds__(256) void prep16v1_kernel(Prep16V1Args a) {
    const V1Geom g = a.g;
    const float qscale = 1.0f / sqrtf((float)g.dk);
    constexpr int KP = F16_KP;
    const long n1 = (long)g.n_head_tiles * 32 * KP, n2 = (long)g.hv * 32 * KP, n3 = (long)F16_QP * F16_DP, n4 = g.hv * 32, n5 = F16_QP;
    const long total = n1 + n2 + n3 + n4 + 2 * n5;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (i < n1) {
            // head tiles: tile = 6 head + {Q0, K0, Q1, K1, V0, V1}
            const int k = (int)(i % KP);
            const long r = i / KP;
            const int f = (int)(r & 31), tile = (int)(r >> 5), head = tile / 6, t6 = tile - 6 * head;
            const int which = t6 < 4 ? (t6 & 1) : 2, blk = t6 < 4 ? (t6 >> 1) : (t6 - 4);
            const int fh = 32 * blk + f;
            float v = 0.f;
            if (fh < g.dk && k < g.d) v = a.w_qkv[((long)which * g.d + head * g.dk + fh) * g.d + k] * (which == 0 ? qscale : 1.0f);
            if (fh < g.dk && k == g.d) v = a.b_qkv[which * g.d + head * g.dk + fh] * (which == 0 ? qscale : 1.0f);      // x16[:, d] = 1
            a.tiles[(long)tile * 32 * KP + dma_tile_pos(f, k)] = (_Float16)v;
        } else if (i < n1 + n2) {
            // W_O tiles: tile j = output features 30 j .. 30 j + 29; column c = attn feature in attn16's order
            const long jx = i - n1;
            const int c = (int)(jx % KP);
            const long r = jx / KP;
            const int f = (int)(r & 31), j = (int)(r >> 5);
            float v = 0.f;
            if (f < g.dkv) {
                const int o = j * g.dkv + f;
                const int ks = c >> 4, p = c & 15;
                if (c == V1_ONES_COL) v = a.b_o[o];
                else if (ks < 3 * g.h) {
                    const int hd = ks / 3, part = ks - 3 * hd;
                    const int hh = p >> 3, jj = p & 7;
                    const int fh = 16 * part + (((jj >> 2) << 3) | (hh << 2) | (jj & 3));
                    if (fh < g.dk && fh < 48) v = a.w_o[(long)o * g.d + hd * g.dk + fh];
                } else if (ks == g.kl && g.lo > 0 && p < g.lo * g.h) {
                    const int hd = p / g.lo, fh = 48 + p % g.lo;
                    v = a.w_o[(long)o * g.d + hd * g.dk + fh];
                }
            }
            a.tiles[((long)g.n_head_tiles + j) * 32 * KP + dma_tile_pos(f, c)] = (_Float16)v;
        } else if (i < n1 + n2 + n3) {
            // additive tiles over the ten output blocks (as prep16_kernel with h = hv, d_k = dkv)
            const long jx = i - n1 - n2;
            const int p = (int)(jx % F16_DP), qq = (int)(jx / F16_DP);
            const int b16 = p >> 4, t = p & 15, hh = t >> 3, jj = t & 7;
            const int fpad = 16 * b16 + 8 * (jj >> 2) + 4 * hh + (jj & 3);
            const int head = fpad >> 5, f = fpad & 31;
            float v = 0.f;
            if (qq < a.q && f < g.dkv && head < g.hv) v = a.w_add[(long)qq * g.d + head * g.dkv + f];
            a.tiles[((long)g.n_head_tiles + g.hv + (qq >> 5)) * 32 * KP + dma_tile_pos(qq & 31, p)] = (_Float16)v;
        } else if (i < n1 + n2 + n3 + n4) {
            // (boeff32: boeff16v1_kernel)
        } else if (i < n1 + n2 + n3 + n4 + n5) {
            const long jx = i - n1 - n2 - n3 - n4;
            a.badd32[jx] = jx < a.q ? a.b_add[jx] * 2.885390082f : 0.f;
        } else {
            const long jx = i - n1 - n2 - n3 - n4 - n5;
            a.qv32[jx] = jx < a.q ? a.q_vec[jx] : 0.f;
        }
    }
}

// the pre-dropout context row of an all-padding title: W_O b_v + b_o per output block; one wave per output feature
__global__ __launch_bounds__(256) void boeff16v1_kernel(V1Geom g, const float* w_o, const float* b_o, const float* b_qkv, float* boeff32) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // idx = 32 j + f
    if (idx >= g.hv * 32) return;
    const int j = idx >> 5, f = idx & 31;
    float v = 0.f;
    if (f < g.dkv) {
        const int o = j * g.dkv + f;
        for (int c = lane; c < g.d; c += 64) v += w_o[(long)o * g.d + c] * b_qkv[2 * g.d + c];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
        v += b_o[o];
    }
    if (lane == 0) boeff32[idx] = v;
}

struct Fwd16V1Args {
    int n_seq, S, q;
    V1Geom g;
    const _Float16* x16;
    const int* pos;
    const int* n_rows;
    const int64_t* ids;
    const int* order;         // [3][n_seq]: long | all-padding | short
    const int* order_cnt;
    const _Float16* wtiles;
    const float *boeff32, *badd32, *qv32;
    _Float16* attn16;         // [n_seq][20 k-steps][32][16]: the head concatenation (v1_attn_col order), kept for d(W_O)
    _Float16* ctx16;          // [n_seq][20][32][16]: W_O output after dropout, v0's context layout
    _Float16* t16;
    float* w;
    float* out;
    Dropout drop;
};

template <bool TRAIN>
__global__ __launch_bounds__(F16_THREADS, 2) void fused_fwd16v1_kernel(Fwd16V1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int S = a.S;
    const V1Geom g = a.g;
    constexpr int KP = F16_KP, DP = F16_DP;
    constexpr float NEG = -3.0e38f;
    const h8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- titles of this wave (as fused_fwd16p_kernel)
    const int n_long = a.order_cnt[0], n_e = a.order_cnt[1], n_short = a.order_cnt[2];
    const int g_pair = (n_short + 2 * F16_WAVES - 1) / (2 * F16_WAVES);
    const int g_long = (n_long + F16_WAVES - 1) / F16_WAVES, g_e = (n_e + F16_WAVES - 1) / F16_WAVES;
    const int blk = blockIdx.x;
    if (blk >= g_pair + g_long + g_e) return;
    const bool pair = blk < g_pair, skip_heads = blk >= g_pair + g_long;
    const int NT = pair ? 2 : 1;
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
    int nlive[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        tok0[i] = (long)seq[i] * S;
        nlive[i] = 0;
        if (pair && val[i]) nlive[i] = __popcll(__ballot(lane < S && a.ids[tok0[i] + lane] != 0));
    }
    const int myp = pair ? (l32 >> 4) : 0, myr = pair ? (l32 & 15) : l32;
    const int myn = myp ? nlive[1] : nlive[0];
    const bool myval = myp ? val[1] : val[0];
    long xrow = -1;
    if (!skip_heads && myval) {
        const long t0 = myp ? tok0[1] : tok0[0];
        if (pair) {
            if (myr < myn) xrow = a.pos[t0 + myr];
            else if (myr == myn && myn < S) xrow = *a.n_rows;
        } else if (myr < S) {
            xrow = a.pos[t0 + myr];
            if (xrow < 0) xrow = *a.n_rows;
        }
    }
    f32x16 kbias;
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
    int src_lane[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) src_lane[i] = 4 * (16 * i + min(l32, nlive[i]) + 32 * hh);
    bool tok_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) tok_ok[i] = val[i] && l32 < S;

    // ---- tile steps: 6 h head tiles and the hv W_O tiles, once per wave (not for all-padding titles), then the 7 additive tiles
    // once per title
    const int n_head_steps = skip_heads ? 0 : g.n_head_tiles + g.hv;
    const int n_steps = n_head_steps + F16_QT * NT;
    auto tile_at = [&](int step) { return step < n_head_steps ? step : g.n_head_tiles + g.hv + (step - n_head_steps) % F16_QT; };
    using Ring = TileRingDMA<3>;
    constexpr int AH = Ring::AHEAD;
    Ring ring;
    ring.smem = smem; ring.src = a.wtiles; ring.n_tiles = n_steps; ring.wave = wave; ring.lane = lane; ring.l32 = l32; ring.hh = hh;
#pragma unroll
    for (int i = 0; i < AH; ++i) ring.load_at(i, tile_at(i));
    // the additive stage's per-row vectors in LDS behind the ring (see fused_fwd16p_kernel)
    float* addv = reinterpret_cast<float*>(smem + 3 * F16_SLOT_DMA);          // [2][F16_QP]
    for (int i = tid; i < 2 * F16_QP; i += F16_THREADS) addv[i] = i < F16_QP ? a.badd32[i] : a.qv32[i - F16_QP];

    h8 xf[F16_KS];
    {
        const _Float16* xr = a.x16 + (xrow < 0 ? 0 : xrow) * KP + 8 * hh;
#pragma unroll
        for (int s = 0; s < F16_KS; ++s) xf[s] = *reinterpret_cast<const h8*>(xr + 16 * s);
        if (xrow < 0) {
#pragma unroll
            for (int s = 0; s < F16_KS; ++s) xf[s] = z8;
        }
    }
    // attn16 holds the TILE's rows: a long title its S token rows; a short title its n real tokens and, in row n, the one row all
    // its padding tokens share (rows beyond: zeros) -- the form the backward's d(W_O) product contracts with the compressed d(o).
    // row_ok: this lane's tile row is such a row.  The k-steps behind the heads' full ones (leftovers, unused, the ones column)
    // start as zeros + the 1.0, and so do the rows 16 .. 31 of a short title's block.
    const int myseq = myp ? seq[1] : seq[0];
    const bool row_ok = myval && (pair ? (myr < myn || (myr == myn && myn < S)) : myr < S);
    if (!skip_heads && myval) {
        for (int ks = g.kl; ks < F16_CS; ++ks) {
            h8 v = z8;
            if (ks == 19 && hh == 0 && row_ok) v[0] = (_Float16)1.0f;
            *reinterpret_cast<h8*>(a.attn16 + frag_off((long)myseq, F16_CS, ks, myr, hh)) = v;
        }
        if (TRAIN && pair)                          // (read by the backward's d(W_O) product only)
            for (int ks = 0; ks < F16_CS; ++ks) *reinterpret_cast<h8*>(a.attn16 + frag_off((long)myseq, F16_CS, ks, 16 + myr, hh)) = z8;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): see fused_fwd16_kernel
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const bool any_live = val[0] || val[1];
    int n = 0;
    auto pre = [&](int gq) { if (n + AH < n_steps) ring.load_piece_at(n + AH, tile_at(n + AH), gq); };
    auto idle_step = [&]() { if (n + AH < n_steps) ring.load_at(n + AH, tile_at(n + AH)); };
    // output block j of title i (its S token rows in the lanes): dropout, ctx16
    auto drop_store = [&](int i, int j, f32x16& cx) {
        if (a.drop.thresh != 0u) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint64_t e0 = (uint64_t)(tok0[i] + l32) * (uint64_t)DP + (uint64_t)(j * 32 + 16 * c + 8 * hh);
                float sc[8];
                dropout_scale8(a.drop.seed, 1u, e0 >> 3, a.drop.thresh16, a.drop.inv_keep, sc);
#pragma unroll
                for (int e = 0; e < 8; ++e) cx[8 * c + e] *= sc[e];
            }
        }
        if (val[i]) {
            _Float16* dst = a.ctx16 + frag_off((long)seq[i], F16_CS, 2 * j, l32, hh);
            *reinterpret_cast<h8*>(dst) = tok_ok[i] ? acc_frag(cx, 0) : z8;
            *reinterpret_cast<h8*>(dst + 512) = tok_ok[i] ? acc_frag(cx, 1) : z8;
        }
    };
    if (!skip_heads) {
#pragma unroll 1
        for (int head = 0; head < g.h; ++head) {
            f32x16 st = zero16();
            // ---- S^T = K0 Q0^T + K1 Q1^T over the two feature blocks
#pragma unroll 1
            for (int b = 0; b < 2; ++b) {
                f32x16 qt = zero16(), kt = zero16();
                if (any_live) tile_mma<true>(qt, ring, n, xf, pre); else idle_step();
                ring.step_barrier(n); ++n;
                if (any_live) tile_mma<true>(kt, ring, n, xf, pre); else idle_step();
                ring.step_barrier(n); ++n;
                if (any_live) {
                    st = mfma32h(acc_frag(kt, 0), acc_frag(qt, 0), st);
                    st = mfma32h(acc_frag(kt, 1), acc_frag(qt, 1), st);
                }
            }
            if (any_live) {
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
            }
            const h8 pf0 = acc_frag(st, 0), pf1 = acc_frag(st, 1);
            // ---- ctx^T of the two blocks: this lane's tile row of it into attn16
#pragma unroll 1
            for (int b = 0; b < 2; ++b) {
                f32x16 vv = zero16(), ct = zero16();
                if (any_live) tile_mma<false>(vv, ring, n, xf, pre); else idle_step();
                if (any_live) {
                    ct = mfma32h(acc_frag(vv, 0), pf0, zero16());
                    ct = mfma32h(acc_frag(vv, 1), pf1, ct);
                }
                if (myval) {
                    _Float16* dst = a.attn16 + frag_off((long)myseq, F16_CS, 3 * head + 2 * b, myr, hh);
                    *reinterpret_cast<h8*>(dst) = row_ok ? acc_frag(ct, 0) : z8;                      // features 32 b .. 32 b + 15
                    if (b == 0) {
                        *reinterpret_cast<h8*>(dst + 512) = row_ok ? acc_frag(ct, 1) : z8;            // features 16 .. 31
                    } else if (g.lo > 0 && hh == 0 && row_ok) {
                        // features 48, 49 (rows 16, 17 of the block: registers 8, 9 of the lower lane half) into the leftover k-step
                        _Float16* lp = a.attn16 + ((long)myseq * F16_CS + g.kl) * 512 + myr * 16 + g.lo * head;
                        for (int e = 0; e < g.lo; ++e) lp[e] = (_Float16)ct[8 + e];
                    }
                }
                ring.step_barrier(n); ++n;
            }
        }
        // ---- W_O on the tile's rows (a pair: both titles at once), then per title its S token rows (token t of a short title = its
        // row min(t, n)), the dropout, ctx16 in v0's layout
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's attn16 stores have landed
        h8 af[F16_CS];
        {
            const _Float16* src = a.attn16 + frag_off((long)myseq, F16_CS, 0, myr, hh);
#pragma unroll
            for (int s = 0; s < F16_CS; ++s) af[s] = *reinterpret_cast<const h8*>(src + 512 * s);
            if (!myval) {
#pragma unroll
                for (int s = 0; s < F16_CS; ++s) af[s] = z8;
            }
        }
#pragma unroll 1
        for (int j = 0; j < g.hv; ++j) {
            f32x16 ot = zero16();
            if (any_live) tile_mma<true>(ot, ring, n, af, pre); else idle_step();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i < NT) {
                    f32x16 cx = ot;
                    if (pair) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            cx[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane[i], __float_as_int(ot[r])));
                    }
                    drop_store(i, j, cx);
                }
            }
            ring.step_barrier(n); ++n;
        }
    } else {
#pragma unroll 1
        for (int j = 0; j < g.hv; ++j) {
            f32x16 cx = rows_of(a.boeff32 + 32 * j, hh);            // all-padding title: W_O b_v + b_o in every row
            drop_store(0, j, cx);
        }
    }

    // ---- per title: the additive attention and the pooling
#pragma unroll 1
    for (int i = 0; i < NT; ++i) {
        const bool valid = i ? val[1] : val[0];
        const bool tok = i ? tok_ok[1] : tok_ok[0];
        const int sq = i ? seq[1] : seq[0];
        const long t0 = i ? tok0[1] : tok0[0];
        // ---- additive attention + pooling over the ten output blocks (fused_fwd16_kernel's last phase)
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
            f32x16 tt = zero16();
            if (valid) tile_mma<true>(tt, ring, n, cf, pre); else idle_step();
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                h4 th;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ex = __builtin_amdgcn_exp2f(fmaf(tt[4 * gq + e], 2.885390082f, ba[4 * gq + e]));
                    const float v = fmaf(-2.0f, __builtin_amdgcn_rcpf(ex + 1.0f), 1.0f);
                    score += qq[4 * gq + e] * v;
                    th[e] = (_Float16)v;
                }
                if (TRAIN && valid) {
                    if (!tok) th = h4{0, 0, 0, 0};
                    *reinterpret_cast<h4*>(a.t16 + (((long)sq * (F16_QP / 16) + 2 * t + (gq >> 1)) * 32 + l32) * 16 + 8 * (gq & 1) + 4 * hh) = th;
                }
            }
            ring.step_barrier(n); ++n;
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
                if (hh == 0 && f < g.dkv && p < g.hv) a.out[(long)sq * g.d + p * g.dkv + f] = acc;
            }
        }
    }
}

int launch_prep16v1(int d, int h, int q, const float* w_qkv, const float* b_qkv, const float* w_o, const float* b_o,
                    const float* w_add, const float* b_add, const float* q_vec, void* planes, hipStream_t stream) {
    const Fused16V1Layout L = fused16v1_layout(d, h, q);
    char* base = (char*)planes;
    Prep16V1Args a{};
    a.g = v1_geom(d, h); a.q = q;
    a.w_qkv = w_qkv; a.b_qkv = b_qkv; a.w_o = w_o; a.b_o = b_o; a.w_add = w_add; a.b_add = b_add; a.q_vec = q_vec;
    a.tiles = (_Float16*)(base + L.tiles); a.boeff32 = (float*)(base + L.boeff32);
    a.badd32 = (float*)(base + L.badd32); a.qv32 = (float*)(base + L.qv32);
    TimingScope ts("prep16", stream);
    hipLaunchKernelGGL(prep16v1_kernel, dim3(2048), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(boeff16v1_kernel, dim3(a.g.hv * 8), dim3(256), 0, stream, a.g, w_o, b_o, b_qkv, a.boeff32);
    return check_launch("prep16v1");
}

int launch_fused_fwd16v1(const Fused16Fwd& f, int h, void* attn16, hipStream_t stream) {
    if (f.n_seq <= 0) return NRMS_OK;
    if (f.order == nullptr || f.pos == nullptr || f.ids == nullptr || f.S > 32) {
        set_error("fused_fwd16v1: needs the padding-skipping path (order lists) and seq_len <= 32");
        return NRMS_EINVAL;
    }
    const Fused16V1Layout L = fused16v1_layout(f.d, h, f.q);
    const char* base = (const char*)f.planes;
    Fwd16V1Args a{};
    a.n_seq = f.n_seq; a.S = f.S; a.q = f.q; a.g = v1_geom(f.d, h);
    a.x16 = (const _Float16*)f.x16; a.pos = f.pos; a.n_rows = f.n_rows; a.ids = f.ids; a.order = f.order; a.order_cnt = f.order_cnt;
    a.wtiles = (const _Float16*)(base + L.tiles); a.boeff32 = (const float*)(base + L.boeff32);
    a.badd32 = (const float*)(base + L.badd32); a.qv32 = (const float*)(base + L.qv32);
    a.attn16 = (_Float16*)attn16; a.ctx16 = (_Float16*)f.ctx16; a.t16 = (_Float16*)f.t16; a.w = f.w; a.out = f.out; a.drop = f.drop;
    const bool train = f.t16 != nullptr;
    const size_t lds = (size_t)3 * F16_SLOT_DMA + (size_t)2 * F16_QP * 4;
    const void* fn = train ? (const void*)fused_fwd16v1_kernel<true> : (const void*)fused_fwd16v1_kernel<false>;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("fused_fwd16v1: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    TimingScope ts("fused_fwd16", stream);
    const dim3 grid(cdiv(f.n_seq, F16_WAVES) + 3);
    if (train) hipLaunchKernelGGL((fused_fwd16v1_kernel<true>), grid, dim3(F16_THREADS), lds, stream, a);
    else hipLaunchKernelGGL((fused_fwd16v1_kernel<false>), grid, dim3(F16_THREADS), lds, stream, a);
    return check_launch("fused_fwd16v1");
}

size_t fused16v1_planes_bytes(int d, int h, int q) { return fused16v1_layout(d, h, q).total; }

}  // namespace nrms
