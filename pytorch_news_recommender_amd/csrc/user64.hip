// Split-bf16 ("bf16x3") user encoder as ONE kernel per direction: sequences of 33..64 rows (the 50-slot click
// histories), d_model <= 300, heads of at most 32 columns, no mask, no dropout -- UserEncoder.forward
// (model/nrms_v0.py:188-199: MultiHeadSelfAttention :46-76 + AdditiveAttention :100-126) and its autograd.
//
// Round 3 ran these 3 % of the step's flops as thirteen latency-bound launches over 25 600 rows (0.57 ms of a 3.3 ms
// step: Q|K|V projection GEMM, attention, additive attention, and their backward chain through fp32 activations in HBM).
// Here a workgroup owns TWO users, a wave one 32-row half of a history; Q, K, V, the attention probabilities and tanh(.)
// live in registers exactly as in the fp16 kernels (fused16.hip: every product's contraction index is where the previous
// product left it), the weight tiles come through an LDS-DMA ring, and the two halves of a history meet in LDS (keys and
// values of the partner block in the forward, the partner's share of dK / dV in the backward).
//
// Arithmetic: every operand is x = hi + lo with hi = bf16(x), lo = bf16(x - hi), every product hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2^-16 relative per product: the precision of the NRMS_PRECISION_BF16X3
// GEMMs this replaces -- here the attention products run in it too instead of on the f32-input MFMA); softmaxes, tanh and
// all HBM activations the remaining GEMMs read (ctx, T, w, dQKV, ds) are fp32.
//
//   forward : x [B,S,d] -> out [B,d];  saves ctx [M,d], T [M,q], w [M] (fp32, row-major: what the weight-gradient GEMMs of
//             the unfused backward read), the Q^T | K^T | V^T operand fragments of every (user, head, block) and the context
//             in operand-fragment order (bf16 hi/lo planes; acts.qkv, nrms_encoder_fused_qkv_bytes)
//   backward: dout [B,d] -> ds [M], d(q_vec) partial sums, dQKV [M,3d] (fp32, head-major -- what d(W_qkv), dX and the bias
//             gradients are computed from by the existing GEMMs)
#include "gemm.h"

namespace nrms {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
struct B2 { b8 hi, lo; };                    // a split operand fragment

#define U64_DI __device__ __forceinline__
// Timing experiments ("what does the kernel cost without this part?") are compiled in only with -DNRMS_U64_EXPERIMENTS and then
// switched by the NRMS_U64_DBG environment variable; they make results WRONG, so a normal build has none of them.
#ifdef NRMS_U64_EXPERIMENTS
#include <stdlib.h>
#define U64_DBG(flags, bit) ((flags) & (bit))
#else
#define U64_DBG(flags, bit) 0
#endif

U64_DI f32x16 mfb(const b8& a, const b8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// acc += a b in split-bf16: the two small terms first
U64_DI void mma3(f32x16& acc, const B2& a, const B2& b) {
    acc = mfb(a.lo, b.hi, acc);
    acc = mfb(a.hi, b.lo, acc);
    acc = mfb(a.hi, b.hi, acc);
}
U64_DI f32x16 zero16u() {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 0.f;
    return r;
}
U64_DI B2 split8(const float (&v)[8]) {
    B2 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        r.hi[j] = h;
        r.lo[j] = (__bf16)(v[j] - (float)h);
    }
    return r;
}
// registers 8s..8s+7 of a 32x32 accumulator as the operand of k-step s (rows of the accumulator = k)
U64_DI B2 acc_frag2(const f32x16& x, int s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[8 * s + j];
    return split8(v);
}
// the same when the accumulator holds bf16-representable values (a transposed plane): one exact conversion
U64_DI b8 acc_plane(const f32x16& x, int s) {
    b8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)x[8 * s + j];
    return r;
}
// X^T from the two operand fragments of X (rows of X = k), plane by plane: a product with the k-permuted identity is exact
// on bf16 values, so the transposed hi / lo planes ARE the split of X^T
U64_DI void transpose2(const B2& x0, const B2& x1, const b8 (&idf)[2], B2 (&out)[2]) {
    f32x16 th = mfb(x0.hi, idf[0], zero16u());
    th = mfb(x1.hi, idf[1], th);
    f32x16 tl = mfb(x0.lo, idf[0], zero16u());
    tl = mfb(x1.lo, idf[1], tl);
#pragma unroll
    for (int s = 0; s < 2; ++s) { out[s].hi = acc_plane(th, s); out[s].lo = acc_plane(tl, s); }
}
U64_DI float regsum16(const f32x16& x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) { s0 += x[4 * g]; s1 += x[4 * g + 1]; s2 += x[4 * g + 2]; s3 += x[4 * g + 3]; }
    return (s0 + s1) + (s2 + s3);
}

constexpr int U_THREADS = 256;              // 4 waves = 2 users x 2 row blocks
constexpr int U_THREADS_C = 256;
constexpr int U_KS = 19;                    // k-steps of the input: 304 >= d + 1 (the ones column carries the Q|K|V bias)
constexpr int U_CS = 20;                    // context k-steps: 10 heads x 32 padded features
constexpr int U_QT = 7;                     // additive tiles of 32: QP = 224 >= q
constexpr int U_QP = 32 * U_QT;
constexpr int U_ZS = U_QP / 16;             // 14 k-steps of dZ
constexpr int U_TILE_KS = 20;               // k-steps a ring slot holds
constexpr int U_TILE = U_TILE_KS * 2048;    // bytes per weight tile: [k-step][plane hi|lo][lane half][row 0..31][8] bf16
constexpr int U_BTILE = U_ZS * 2048;        // backward tile (Wadd_h^T: 32 features x 224 q)
constexpr float U_NEG = -3.0e38f;

// ---- weight-tile ring: tile n lives in LDS slot n & 1; while tile n is consumed, tile n + 1 travels global -> registers (requested
// at the top of the step) -> LDS (copied at its end, in front of the step's barrier).  NOT LDS-DMA: measured on this kernel,
// global_load_lds took in 28 GB/s per CU (37 tiles of 40 KB in 52 us with nothing else running, the same for 32 or 256
// workgroups: the issuing wave stalls on every piece), as long as a step's 57 MFMAs take -- and with one wave per SIMD nobody
// covers the stall.  Plain 16-byte loads issue in a few cycles, land in registers while the MFMAs run, and cost 40 of the
// kernel's 512 registers.  Measured and dropped: the copy and the next request interleaved chunk by chunk between the MFMA
// groups, with one register set (forward 0.123 -> 0.165 ms) or two (0.152 ms) -- a vector-memory wait in the middle of an MFMA
// group costs more than the ~0.3 us per step that the block copy at the step's end leaves uncovered.
// The tile is stored in HBM in the order its fragments are read ([k-step][plane][lane half][row][8]): the copy is
// thread-linear and a wave's fragment read is one contiguous KiB (conflict-free).
template <int TILE_BYTES>
struct URing {
    static constexpr int CH = TILE_BYTES / (U_THREADS_C * 16);      // 16-byte chunks per thread and tile
    static_assert(CH * U_THREADS_C * 16 == TILE_BYTES, "whole chunks");
    char* smem;
    const char* src;
    int n_tiles, tid, l32, hh;
    b8 stg[CH];
    U64_DI void load(int n) {
        if (n >= n_tiles) return;
        const b8* g = reinterpret_cast<const b8*>(src + (long)n * TILE_BYTES) + tid;
#pragma unroll
        for (int i = 0; i < CH; ++i) stg[i] = g[i * U_THREADS_C];
    }
    U64_DI void store(int n) {
        if (n >= n_tiles) return;
        b8* l = reinterpret_cast<b8*>(smem + (n & 1) * TILE_BYTES) + tid;
#pragma unroll
        for (int i = 0; i < CH; ++i) l[i * U_THREADS_C] = stg[i];
    }
    // end of a step: this wave's LDS traffic (fragment reads of tile n, the copy of tile n + 1) is complete, then the workgroup meets
    U64_DI void step_barrier() const { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    U64_DI unsigned frag_base(int n) const { return (unsigned)(uintptr_t)(smem + (n & 1) * TILE_BYTES) + (unsigned)((hh * 32 + l32) * 16); }
};

// A fragment read the compiler does not track: issued here, complete only after a frag_wait that names it.  (hipcc answers a
// tracked prefetch with s_waitcnt lgkmcnt(0) in every second MFMA group -- a full LDS round trip on the reads it has just issued.)
template <int OFF>
U64_DI b8 frag_async(unsigned addr) {
    b8 r;
    __asm__ volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// everything but the N newest LDS operations of this wave is complete; the "+v" operands tie the consumers of w0, w1 behind it
template <int N>
U64_DI void frag_wait(B2& w0, B2& w1) {
    __asm__ volatile("s_waitcnt lgkmcnt(%4)" : "+v"(w0.hi), "+v"(w0.lo), "+v"(w1.hi), "+v"(w1.lo) : "n"(N));
}

// acc += (W tile n) x (register operand) over NS k-steps, two k-steps per group: the weight fragments of group g + 1 are read
// from LDS while the six MFMAs of group g issue.  W_IS_A: acc = W op^T (features x tokens); else acc = op W^T.
template <int NS, bool W_IS_A, class Ring, int NOP>
U64_DI void tile_mma3(f32x16& acc, const Ring& ring, int n, const B2 (&op)[NOP]) {
    static_assert(NS <= NOP && NS >= 2, "operand k-steps");
    constexpr int NG = (NS + 1) / 2;
    const unsigned base = ring.frag_base(n);
    B2 wa[2], wb[2];
    wa[0].hi = frag_async<0>(base); wa[0].lo = frag_async<1024>(base);
    wa[1].hi = frag_async<2048>(base); wa[1].lo = frag_async<3072>(base);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const bool more = g + 1 < NG;
        if (more) {
            const unsigned b2 = base + 4096 * (g + 1);
            B2(&nx)[2] = (g & 1) ? wa : wb;
            nx[0].hi = frag_async<0>(b2); nx[0].lo = frag_async<1024>(b2);
            if (2 * (g + 1) + 1 < NS) { nx[1].hi = frag_async<2048>(b2); nx[1].lo = frag_async<3072>(b2); }
        }
        B2(&w)[2] = (g & 1) ? wb : wa;
        // the reads of group g + 1 (4, or 2 for an odd tail) may stay in flight
        if (!more) frag_wait<0>(w[0], w[1]);
        else if (2 * (g + 1) + 1 < NS) frag_wait<4>(w[0], w[1]);
        else frag_wait<2>(w[0], w[1]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = 2 * g + i;
            if (s < NS) {
                if (W_IS_A) mma3(acc, w[i], op[s]); else mma3(acc, op[s], w[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// padded context feature held at position p of its row (P16 order inside every 16-block: fused16.hip) -> (head, feature)
U64_DI int p16_feature(int p) {
    const int b16 = p >> 4, t = p & 15, hh = t >> 3, jj = t & 7;
    return 16 * b16 + 8 * (jj >> 2) + 4 * hh + (jj & 3);
}

// S^T[j][i] of one query block against both key blocks -> P^T (softmax over the keys = the rows: registers, the lane-half
// exchange and the two key blocks), identical instruction sequence in the forward and in the backward's recomputation
// Key blocks are indexed [0] = the wave's OWN 32 rows, [1] = the partner's (no register array is indexed by the runtime block
// number); blk = the wave's block, so block jb starts at token 32 (blk ^ jb).
U64_DI void attn_probs(const B2 (&kf)[2][2], const B2 (&qf)[2], int S, int blk, int hh, f32x16 (&pt)[2]) {
    float m = U_NEG;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        pt[jb] = zero16u();
        mma3(pt[jb], kf[jb][0], qf[0]);
        mma3(pt[jb], kf[jb][1], qf[1]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pt[jb][r] = 32 * (blk ^ jb) + crow32(r, hh) < S ? pt[jb][r] : U_NEG;       // rows beyond the sequence are not keys
            m = fmaxf(m, pt[jb][r]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = 32 * (blk ^ jb) + crow32(r, hh) < S ? __expf(pt[jb][r] - m) : 0.f;
            pt[jb][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[jb][r] *= inv;
}

struct U64FwdArgs {
    int n_seq, S, d, h, dk, q;
    const float* x;          // [n_seq, S, d]
    const char* wtiles;      // [3h + 7] tiles of U_TILE bytes: per head W'_q | W_k | W_v (Q pre-scaled by 1/sqrt(d_k)), then the additive tiles
    const float* addv;       // [2][U_QP]: b_add, q_vec (zero padded)
    float* ctx;              // [M, d] or null (inference)
    float* t;                // [M, q] or null
    float* w;                // [M] or null
    b8* qkvf;                // [n_seq][h][2 blocks][3][2 k-steps][2 planes][64] operand fragments, or null (inference)
    b8* ctxf;                // [n_seq][2 blocks][U_CS][2 planes][64]: the context in operand order (always: re-read by the additive stage)
    float* out;              // [n_seq, d]
    int dbg;                 // experiments only: 1 no head-tile MFMAs, 2 no attention, 4 no training stores, 8 no additive tiles, 16 no x loads
    unsigned long long* stamps;   // experiments only: s_memrealtime (100 MHz) of workgroup 0 at phase boundaries
};
#ifdef NRMS_U64_EXPERIMENTS
#define U64_STAMP(a, i) do { if ((a).stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { (a).stamps[i] = __builtin_amdgcn_s_memrealtime(); (a).stamps[32 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define U64_STAMP(a, i) do {} while (0)
#endif

// LDS: ring 2 x U_TILE | exchange [4 waves][8 KiB] | addv [2][U_QP] floats | red [4 waves][4] | pout [4][U_CS * 16] floats
constexpr int U_FWD_EXCH = 2 * U_TILE;
constexpr int U_FWD_ADDV = U_FWD_EXCH + 4 * 8192;
constexpr int U_FWD_RED = U_FWD_ADDV + 2 * U_QP * 4;
constexpr int U_FWD_POUT = U_FWD_RED + 64;
constexpr int U_FWD_LDS = U_FWD_POUT + 4 * U_CS * 16 * 4;

template <bool TRAIN>
__global__ __launch_bounds__(U_THREADS, 1) void user64_fwd_kernel(U64FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int S = a.S, blk = wave & 1;
    const int seq_raw = blockIdx.x * 2 + (wave >> 1);
    const bool valid = seq_raw < a.n_seq;
    const int seq = valid ? seq_raw : 0;
    const int tok = 32 * blk + l32;
    const bool tok_ok = valid && tok < S;
    const long row = (long)seq * S + tok;
    const int n_head_tiles = 3 * a.h;

    using Ring = URing<U_TILE>;
    Ring ring;
    ring.smem = smem; ring.src = a.wtiles; ring.n_tiles = n_head_tiles + U_QT; ring.tid = tid; ring.l32 = l32; ring.hh = hh;
    ring.load(0);
    float* addv = reinterpret_cast<float*>(smem + U_FWD_ADDV);
    for (int i = tid; i < 2 * U_QP; i += U_THREADS) addv[i] = a.addv[i];

    // ---- this lane's input fragments: token tok, features 16 s + 8 hh .. + 7, split; column d is the ones column
    B2 xf[U_KS];
    {
        // unconditional loads from clamped addresses (a predicated load is a branch and a wait each), the values selected afterwards
        const float* xr = a.x + (tok_ok && !U64_DBG(a.dbg, 16) ? row : 0) * a.d;
#pragma unroll
        for (int s = 0; s < U_KS; ++s) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int f0 = 16 * s + 8 * hh + 4 * c;
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(xr + (f0 < a.d ? f0 : 0));        // d % 4 == 0
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * c + e] = !tok_ok ? 0.f : (f0 < a.d ? t4[e] : ((f0 == a.d && e == 0) ? 1.0f : 0.f));
            }
            xf[s] = split8(v);
        }
    }
    b8 idf[2];                                  // k-permuted identity (transposing products)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) idf[s][j] = (__bf16)(l32 == 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3) ? 1.0f : 0.0f);
    U64_STAMP(a, 0);
    ring.store(0);
    ring.step_barrier();                        // tile 0 and addv are in LDS
    U64_STAMP(a, 1);

    char* exch = smem + U_FWD_EXCH;
    int n = 0;
#pragma unroll 1
    for (int head = 0; head < a.h; ++head) {
        f32x16 qt = zero16u(), kt = zero16u(), vv = zero16u();
        if (head == 1) U64_STAMP(a, 6);
        if (head == 2) U64_STAMP(a, 7);
        ring.load(n + 1);
        if (!U64_DBG(a.dbg, 1)) tile_mma3<U_KS, true>(qt, ring, n, xf);
        ring.store(n + 1);
        ring.step_barrier();
        ++n;
        ring.load(n + 1);
        if (!U64_DBG(a.dbg, 1)) tile_mma3<U_KS, true>(kt, ring, n, xf);
        ring.store(n + 1);
        ring.step_barrier();
        ++n;
        ring.load(n + 1);
        if (!U64_DBG(a.dbg, 1)) tile_mma3<U_KS, false>(vv, ring, n, xf);
        ring.store(n + 1);
        // operand fragments of this block: Q^T, K^T (k = feature), V (k = token)
        B2 qf[2], kf[2][2], vf[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = acc_frag2(qt, s);
            kf[0][s] = acc_frag2(kt, s);              // [0] = this block, [1] = the partner's (attn_probs)
            vf[0][s] = acc_frag2(vv, s);
        }
        // the partner wave (the other 32 rows of this history) needs K and V of this block as operands
        {
            b8* mine = reinterpret_cast<b8*>(exch + wave * 8192) + lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                mine[(0 + 2 * s) * 64] = kf[0][s].hi; mine[(1 + 2 * s) * 64] = kf[0][s].lo;
                mine[(4 + 2 * s) * 64] = vf[0][s].hi; mine[(5 + 2 * s) * 64] = vf[0][s].lo;
            }
        }
        ring.step_barrier();                    // (also: the exchange buffer is complete)
        ++n;
        if (head == 1) U64_STAMP(a, 8);
        if (TRAIN && valid && !U64_DBG(a.dbg, 4)) {
            // saved for the backward: Q^T, K^T and V^T fragments (all with k = feature: the transposed V planes)
            B2 vt[2];
            transpose2(vf[0][0], vf[0][1], idf, vt);
            b8* dst = a.qkvf + ((((long)seq * a.h + head) * 2 + blk) * 12) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dst[(0 + 2 * s) * 64] = qf[s].hi; dst[(1 + 2 * s) * 64] = qf[s].lo;
                dst[(4 + 2 * s) * 64] = kf[0][s].hi; dst[(5 + 2 * s) * 64] = kf[0][s].lo;
                dst[(8 + 2 * s) * 64] = vt[s].hi; dst[(9 + 2 * s) * 64] = vt[s].lo;
            }
        }
        {
            const b8* theirs = reinterpret_cast<const b8*>(exch + (wave ^ 1) * 8192) + lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kf[1][s].hi = theirs[(0 + 2 * s) * 64]; kf[1][s].lo = theirs[(1 + 2 * s) * 64];
                vf[1][s].hi = theirs[(4 + 2 * s) * 64]; vf[1][s].lo = theirs[(5 + 2 * s) * 64];
            }
        }
        // ---- attention of this head for the 32 queries of this block, in registers
        f32x16 pt[2];
        f32x16 ct = zero16u();                  // ctx^T[f][i] = sum_j V[j][f] P^T[j][i]
        if (!U64_DBG(a.dbg, 2)) {
        attn_probs(kf, qf, S, blk, hh, pt);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int s = 0; s < 2; ++s) mma3(ct, vf[jb][s], acc_frag2(pt[jb], s));
        } else { ct = qt; }
        if (valid) {
            b8* dst = a.ctxf + (((long)seq * 2 + blk) * U_CS + 2 * head) * 128 + lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const B2 c2 = acc_frag2(ct, s);
                dst[(2 * s) * 64] = c2.hi;
                dst[(2 * s + 1) * 64] = c2.lo;
            }
            if (TRAIN && tok_ok && !U64_DBG(a.dbg, 4)) {
                float* crow = a.ctx + row * a.d + head * a.dk;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int f = 8 * g + 4 * hh + e;
                        if (f < a.dk) { crow[f] = ct[4 * g + e]; crow[f + 1] = ct[4 * g + e + 1]; }      // d_k is even
                    }
            }
        }
    }
    U64_STAMP(a, 2);
    if (valid) {                                // heads the model does not have: zero context columns
        const b8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 2 * a.h; c < U_CS; ++c) {
            b8* dst = a.ctxf + (((long)seq * 2 + blk) * U_CS + c) * 128 + lane;
            dst[0] = z; dst[64] = z;
        }
    }

    // ---- additive attention: T^T[q][tok] = sum_f Wadd[q][f] ctx^T[f][tok]; the context operand is read back in the order it
    // was stored in (own stores: wait for them, then plain loads)
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B2 cf[U_CS];
    {
        const b8* src = a.ctxf + ((long)seq * 2 + blk) * U_CS * 128 + lane;
#pragma unroll
        for (int s = 0; s < U_CS; ++s) { cf[s].hi = src[(2 * s) * 64]; cf[s].lo = src[(2 * s + 1) * 64]; }
    }
    U64_STAMP(a, 3);
    float score = 0.f;
    f32x4 tkeep[4];                             // tanh(.) of a tile: stored at the start of the next step (behind the barrier)
    auto store_t = [&](int t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q0 = 32 * t + 8 * g + 4 * hh;
            if (TRAIN && tok_ok && q0 < a.q && !U64_DBG(a.dbg, 4)) *reinterpret_cast<f32x4*>(a.t + row * a.q + q0) = tkeep[g];      // q % 4 == 0
        }
    };
#pragma unroll 1
    for (int t = 0; t < U_QT; ++t) {
        if (t > 0) store_t(t - 1);
        ring.load(n + 1);
        f32x16 tt = zero16u();
        if (!U64_DBG(a.dbg, 8)) tile_mma3<U_CS, true>(tt, ring, n, cf);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ba = *reinterpret_cast<const f32x4*>(addv + 32 * t + 8 * g + 4 * hh);
            const f32x4 qq = *reinterpret_cast<const f32x4*>(addv + U_QP + 32 * t + 8 * g + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                tkeep[g][e] = fast_tanh(tt[4 * g + e] + ba[e]);
                score += qq[e] * tkeep[g][e];
            }
        }
        ring.store(n + 1);
        ring.step_barrier();
        ++n;
    }
    store_t(U_QT - 1);
    U64_STAMP(a, 4);
    // ---- softmax over the tokens of the history: the lanes of both waves of the pair
    float* red = reinterpret_cast<float*>(smem + U_FWD_RED);       // [4 waves][4]
    score += __shfl_xor(score, 32, 64);
    score = tok < S ? score : U_NEG;
    float mx = score;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave * 4] = mx;
    __syncthreads();
    mx = fmaxf(red[wave * 4], red[(wave ^ 1) * 4]);
    float wgt = tok < S ? __expf(score - mx) : 0.f;
    float es = wgt;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) es += __shfl_xor(es, o, 64);
    if (lane == 0) red[wave * 4 + 1] = es;
    __syncthreads();
    es = red[(wave & 2) * 4 + 1] + red[((wave & 2) + 1) * 4 + 1];   // block 0 + block 1: the same order in both waves
    wgt /= es;
    if (TRAIN && tok_ok && hh == 0) a.w[row] = wgt;

    // ---- pooling: out[f] = sum_tok w_tok ctx[tok][f].  A product with a column selector moves the tokens from the lanes
    // into the accumulator's rows (fused16.hip); hi and lo planes add up to the 16-bit context
    {
        b8 sel[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) sel[s2][j] = (__bf16)(l32 == 16 * s2 + 8 * hh + j ? 1.0f : 0.0f);
        float wrow[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) wrow[r] = __shfl(wgt, crow32(r, hh), 64);
        float* pout = reinterpret_cast<float*>(smem + U_FWD_POUT) + wave * (U_CS * 16);
#pragma unroll
        for (int p = 0; p < U_CS / 2; ++p) {
            f32x16 dd = mfb(cf[2 * p].lo, sel[0], zero16u());
            dd = mfb(cf[2 * p + 1].lo, sel[1], dd);
            dd = mfb(cf[2 * p].hi, sel[0], dd);
            dd = mfb(cf[2 * p + 1].hi, sel[1], dd);
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc += wrow[r] * dd[r];
            acc += __shfl_xor(acc, 32, 64);
            if (hh == 0) pout[32 * p + l32] = acc;
        }
    }
    __syncthreads();
    {
        // column n of head p is P16 position n: padded feature p16_feature(n)
        const float* pout = reinterpret_cast<const float*>(smem + U_FWD_POUT);
        for (int i = tid; i < 2 * U_CS * 16; i += U_THREADS) {
            const int u = i / (U_CS * 16), c = i - u * (U_CS * 16);
            const int p = c >> 5, f = p16_feature(c & 31);
            const int sq = blockIdx.x * 2 + u;
            if (sq < a.n_seq && p < a.h && f < a.dk)
                a.out[(long)sq * a.d + p * a.dk + f] = pout[(2 * u) * (U_CS * 16) + c] + pout[(2 * u + 1) * (U_CS * 16) + c];
        }
    }
    U64_STAMP(a, 5);
}

// ---------------------------------------------------------------------------------------------------------------
struct U64BwdArgs {
    int n_seq, S, d, h, dk, q;
    float qscale;            // 1 / sqrt(d_k): dQ = dQ' * qscale
    const char* btiles;      // [h] tiles of U_BTILE bytes: Wadd_h^T (32 features x 224 q)
    const float* qv;         // [U_QP] q_vec, zero padded
    const float* dout;       // [n_seq, d]
    const float* t;          // [M, q]
    const float* w;          // [M]
    const b8* qkvf;          // forward
    const b8* ctxf;          // forward
    float* ds;               // [M]
    float* dq_partial;       // [gridDim.x][q]: this workgroup's share of d(q_vec)
    float* dqkv;             // [M, 3d] head-major: head h = columns [3 d_k h, 3 d_k (h + 1)) as Q | K | V
};

// LDS: ring 2 x U_BTILE | exchange [2 parities][4 waves][8 KiB] | dout [2][320] floats | red [4][4] | stg [4 waves][7][32] floats
constexpr int U_BWD_EXCH = 2 * U_BTILE;
constexpr int U_BWD_DOUT = U_BWD_EXCH + 2 * 4 * 8192;
constexpr int U_BWD_RED = U_BWD_DOUT + 2 * 320 * 4;
constexpr int U_BWD_STG = U_BWD_RED + 64;
constexpr int U_BWD_LDS = U_BWD_STG + 4 * U_QT * 32 * 4;

__global__ __launch_bounds__(U_THREADS, 1) void user64_bwd_kernel(U64BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    const int S = a.S, blk = wave & 1;
    const int seq_raw = blockIdx.x * 2 + (wave >> 1);
    const bool valid = seq_raw < a.n_seq;
    const int seq = valid ? seq_raw : 0;
    const int tok = 32 * blk + l32;
    const bool tok_ok = valid && tok < S;
    const long row = (long)seq * S + tok;

    using Ring = URing<U_BTILE>;
    Ring ring;
    ring.smem = smem; ring.src = a.btiles; ring.n_tiles = a.h; ring.tid = tid; ring.l32 = l32; ring.hh = hh;
    ring.load(0);
    // dout rows of the two users in the padded [head][32] layout (zeros in the padding columns)
    float* dsh = reinterpret_cast<float*>(smem + U_BWD_DOUT);
    for (int i = tid; i < 2 * 320; i += U_THREADS) {
        const int u = i / 320, c = i - u * 320, head = c >> 5, f = c & 31;
        const int sq = blockIdx.x * 2 + u;
        dsh[i] = (sq < a.n_seq && head < a.h && f < a.dk) ? a.dout[(long)sq * a.d + head * a.dk + f] : 0.f;
    }
    b8 idf[2], sel[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            idf[s][j] = (__bf16)(l32 == 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3) ? 1.0f : 0.0f);
            sel[s][j] = (__bf16)(l32 == 16 * s + 8 * hh + j ? 1.0f : 0.0f);
        }
    const float wgt = tok_ok ? a.w[row] : 0.f;
    __syncthreads();
    const float* dmy = dsh + (wave >> 1) * 320;

    // ================= pooling backward =================
    // dw_tok = <dout, ctx_tok>: A = dout (the same row in every lane), B = the context fragments; all rows of the result equal
    float dw;
    {
        f32x16 acc = zero16u();
        const b8* src = a.ctxf + ((long)seq * 2 + blk) * U_CS * 128 + lane;
#pragma unroll
        for (int s = 0; s < U_CS; ++s) {
            B2 cfs;
            cfs.hi = src[(2 * s) * 64];
            cfs.lo = src[(2 * s + 1) * 64];
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dmy[p16_feature(16 * s + 8 * hh + j)];
            mma3(acc, split8(v), cfs);
        }
        dw = acc[0];
    }
    float* red = reinterpret_cast<float*>(smem + U_BWD_RED);
    float aw = wgt * dw;
    // (both lane halves hold the same 32 tokens: the reduction stays inside a half)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) aw += __shfl_xor(aw, o, 64);
    if (lane == 0) red[wave * 4] = aw;
    __syncthreads();
    aw = red[(wave & 2) * 4] + red[((wave & 2) + 1) * 4];
    const float ds = wgt * (dw - aw);            // 0 for lanes beyond the sequence
    if (tok_ok && hh == 0) a.ds[row] = ds;

    // dZ[tok][q] = ds q_vec[q] (1 - T^2) as operand fragments (k = q), U[tok][q] = ds T (column sums = d(q_vec))
    B2 zf[U_ZS];
    float* stg = reinterpret_cast<float*>(smem + U_BWD_STG);
#pragma unroll
    for (int t = 0; t < U_QT; ++t) {
        f32x16 accu = zero16u();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int s = 2 * t + s2;
            float dz[8], u[8];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int q0 = 16 * s + 8 * hh + 4 * c;
                f32x4 tv = {0.f, 0.f, 0.f, 0.f};
                if (tok_ok && q0 < a.q) tv = *reinterpret_cast<const f32x4*>(a.t + row * a.q + q0);
                const f32x4 qq = *reinterpret_cast<const f32x4*>(a.qv + q0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dz[4 * c + e] = ds * qq[e] * (1.0f - tv[e] * tv[e]);
                    u[4 * c + e] = ds * tv[e];
                }
            }
            zf[s] = split8(dz);
            const B2 uf = split8(u);
            accu = mfb(uf.lo, sel[s2], accu);    // D[tok][n] = U[tok][q = 32 t + n]
            accu = mfb(uf.hi, sel[s2], accu);
        }
        float cu = regsum16(accu);
        cu += __shfl_xor(cu, 32, 64);
        if (hh == 0) stg[(wave * U_QT + t) * 32 + l32] = cu;
    }
    ring.store(0);
    ring.step_barrier();                          // tile 0 in LDS; stg complete
    for (int i = tid; i < a.q; i += U_THREADS) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += stg[w * U_QP + i];            // fixed order: waves ascending
        a.dq_partial[(long)blockIdx.x * a.q + i] = sum;
    }

    // ================= d(ctx)^T per head -> operand fragments, parked where the context's were =================
    // d(ctx)^T[f][tok] = sum_q Wadd[q][f] dZ[tok][q] + w_tok dout[f].  The fragments go to HBM (written and re-read by the same
    // lanes, over the context fragments this wave has just consumed) so that the 112 registers of dZ are dead in the attention loop.
    b8* dcf = const_cast<b8*>(a.ctxf) + ((long)seq * 2 + blk) * U_CS * 128 + lane;
#pragma unroll 1
    for (int head = 0; head < a.h; ++head) {
        f32x16 dct = zero16u();
        ring.load(head + 1);
        tile_mma3<U_ZS, true>(dct, ring, head, zf);
#pragma unroll
        for (int r = 0; r < 16; ++r) dct[r] += wgt * dmy[head * 32 + crow32(r, hh)];
        if (valid) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const B2 c2 = acc_frag2(dct, s);
                dcf[((2 * head + s) * 2) * 64] = c2.hi;
                dcf[((2 * head + s) * 2 + 1) * 64] = c2.lo;
            }
        }
        ring.store(head + 1);
        ring.step_barrier();
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the d(ctx) fragments are re-read below (own stores)

    // ================= the attention backward, head by head =================
    char* exch = smem + U_BWD_EXCH;
    // the forward's operand fragments of a head: Q^T of this block, K^T and V^T of both blocks ([0] = this block, [1] = the
    // partner's); d(ctx)^T of this block.  Head h + 1's are requested while head h is computed (24 loads of 1 KiB per wave: an
    // exposed L2 round trip per head otherwise)
    B2 qf[2], kf[2][2], vtf[2][2], dc[2], nqf[2], nkf[2][2], nvtf[2][2], ndc[2];
    auto fetch = [&](int head, B2 (&q_)[2], B2 (&k_)[2][2], B2 (&v_)[2][2], B2 (&d_)[2]) {
        const b8* base = a.qkvf + (((long)seq * a.h + head) * 2) * 12 * 64 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const b8* mine = base + blk * 12 * 64;
            q_[s].hi = mine[(0 + 2 * s) * 64]; q_[s].lo = mine[(1 + 2 * s) * 64];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const b8* src = base + (blk ^ jb) * 12 * 64;
                k_[jb][s].hi = src[(4 + 2 * s) * 64]; k_[jb][s].lo = src[(5 + 2 * s) * 64];
                v_[jb][s].hi = src[(8 + 2 * s) * 64]; v_[jb][s].lo = src[(9 + 2 * s) * 64];
            }
            d_[s].hi = dcf[((2 * head + s) * 2) * 64];
            d_[s].lo = dcf[((2 * head + s) * 2 + 1) * 64];
        }
    };
    fetch(0, nqf, nkf, nvtf, ndc);
#pragma unroll 1
    for (int head = 0; head < a.h; ++head) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[s] = nqf[s]; dc[s] = ndc[s];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) { kf[jb][s] = nkf[jb][s]; vtf[jb][s] = nvtf[jb][s]; }
        }
        if (head + 1 < a.h) fetch(head + 1, nqf, nkf, nvtf, ndc);
        // ---- P^T recomputed; dP^T[j][i] = sum_f V^T[f][j] d(ctx)^T[f][i]; dS^T = P^T o (dP^T - delta_i)
        f32x16 pt[2], dst[2];
        attn_probs(kf, qf, S, blk, hh, pt);
        float delta = 0.f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            dst[jb] = zero16u();
            mma3(dst[jb], vtf[jb][0], dc[0]);
            mma3(dst[jb], vtf[jb][1], dc[1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) delta += pt[jb][r] * dst[jb][r];
        }
        delta += __shfl_xor(delta, 32, 64);
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[jb][r] = pt[jb][r] * (dst[jb][r] - delta);

        // d(ctx) [i][f] and Q' [i][f] of this block as operands with k = query
        B2 dx0[2], qx[2];
        transpose2(dc[0], dc[1], idf, dx0);
        transpose2(qf[0], qf[1], idf, qx);
        // ---- this block's queries against the keys of block jb: dV^T[f][j] = sum_i d(ctx)[i][f] P[i][j],
        //      dK^T[f][j] = sum_i Q'[i][f] dS[i][j]  (partial sums over the 32 queries of this wave)
        f32x16 dvp[2], dkp[2];
        f32x16 dq = zero16u();
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            B2 pf[2], sf[2], px[2], sx[2], kx[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) { pf[s] = acc_frag2(pt[jb], s); sf[s] = acc_frag2(dst[jb], s); }
            transpose2(pf[0], pf[1], idf, px);            // P[i][j]
            transpose2(sf[0], sf[1], idf, sx);            // dS[i][j]
            dvp[jb] = zero16u();
            dkp[jb] = zero16u();
#pragma unroll
            for (int s = 0; s < 2; ++s) { mma3(dvp[jb], dx0[s], px[s]); mma3(dkp[jb], qx[s], sx[s]); }
            // dQ'^T[f][i] += sum_j K[j][f] dS^T[j][i]
            transpose2(kf[jb][0], kf[jb][1], idf, kx);    // K[j][f]
#pragma unroll
            for (int s = 0; s < 2; ++s) mma3(dq, kx[s], sf[s]);
        }
        // the other block's keys belong to the partner wave: hand over this wave's share of their dK, dV (double buffered
        // over the heads: the partner reads buffer head & 1 behind the barrier while this wave may already fill the other)
        {
            float* mine = reinterpret_cast<float*>(exch + ((head & 1) * 4 + wave) * 8192) + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) { mine[r * 64] = dkp[1][r]; mine[(16 + r) * 64] = dvp[1][r]; }
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const float* theirs = reinterpret_cast<const float*>(exch + ((head & 1) * 4 + (wave ^ 1)) * 8192) + lane;
            // (two terms: the sum is the same bits in either order)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dkp[0][r] += theirs[r * 64]; dvp[0][r] += theirs[(16 + r) * 64]; }
        }
        if (tok_ok) {
            float* orow = a.dqkv + row * (3L * a.d) + 3 * a.dk * head;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int f = 8 * g + 4 * hh + e;
                    if (f < a.dk) {
                        orow[f] = dq[4 * g + e] * a.qscale; orow[f + 1] = dq[4 * g + e + 1] * a.qscale;
                        orow[a.dk + f] = dkp[0][4 * g + e]; orow[a.dk + f + 1] = dkp[0][4 * g + e + 1];
                        orow[2 * a.dk + f] = dvp[0][4 * g + e]; orow[2 * a.dk + f + 1] = dvp[0][4 * g + e + 1];
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight planes: one launch per call
struct U64PrepArgs {
    int d, h, dk, q;
    const float* w_qkv;   // [3d][d]
    const float* b_qkv;   // [3d]
    const float* w_add;   // [q][d]
    const float* b_add;   // [q]
    const float* q_vec;   // [q]
    __bf16* wtiles;       // forward tiles (null: skip)
    float* addv;          // [2][U_QP]
    __bf16* btiles;       // backward tiles (null: skip)
    float* qv;            // [U_QP]
};

// element (row r, k) of a tile: [k-step][plane][lane half][row][8]
__host__ __device__ __forceinline__ long u64_tile_pos(int r, int k, int plane) {
    return ((((long)(k >> 4) * 2 + plane) * 2 + ((k >> 3) & 1)) * 32 + r) * 8 + (k & 7);
}

__global__ __launch_bounds__(256) void user64_prep_kernel(U64PrepArgs a) {
    const float qscale = 1.0f / sqrtf((float)a.dk);
    const long per_tile = 32L * 16 * U_TILE_KS;                        // (row, k) pairs of a forward tile
    const long n_fwd = a.wtiles != nullptr ? (3L * a.h + U_QT) * per_tile : 0;
    const long per_btile = 32L * 16 * U_ZS;
    const long n_bwd = a.btiles != nullptr ? (long)a.h * per_btile : 0;
    const long n_vec = 3L * U_QP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_fwd + n_bwd + n_vec; i += (long)gridDim.x * blockDim.x) {
        if (i < n_fwd) {
            const int tile = (int)(i / per_tile);
            const int rk = (int)(i - (long)tile * per_tile), r = rk / (16 * U_TILE_KS), k = rk - r * (16 * U_TILE_KS);
            float v = 0.f;
            if (tile < 3 * a.h) {
                const int head = tile / 3, which = tile - 3 * head;
                if (r < a.dk && k < a.d) v = a.w_qkv[((long)which * a.d + head * a.dk + r) * a.d + k];
                if (r < a.dk && k == a.d) v = a.b_qkv[which * a.d + head * a.dk + r];          // the ones column of x
                if (which == 0) v *= qscale;
            } else {
                const int qq = 32 * (tile - 3 * a.h) + r, fpad = p16_feature(k), head = fpad >> 5, f = fpad & 31;
                if (qq < a.q && head < a.h && f < a.dk) v = a.w_add[(long)qq * a.d + head * a.dk + f];
            }
            const __bf16 hi = (__bf16)v;
            __bf16* t = a.wtiles + (long)tile * (U_TILE / 2);
            t[u64_tile_pos(r, k, 0)] = hi;
            t[u64_tile_pos(r, k, 1)] = (__bf16)(v - (float)hi);
        } else if (i < n_fwd + n_bwd) {
            const long j = i - n_fwd;
            const int head = (int)(j / per_btile);
            const int rk = (int)(j - (long)head * per_btile), f = rk / (16 * U_ZS), qq = rk - f * (16 * U_ZS);
            float v = 0.f;
            if (f < a.dk && qq < a.q) v = a.w_add[(long)qq * a.d + head * a.dk + f];
            const __bf16 hi = (__bf16)v;
            __bf16* t = a.btiles + (long)head * (U_BTILE / 2);
            t[u64_tile_pos(f, qq, 0)] = hi;
            t[u64_tile_pos(f, qq, 1)] = (__bf16)(v - (float)hi);
        } else {
            const int j = (int)(i - n_fwd - n_bwd), which = j / U_QP, qq = j - which * U_QP;
            if (which == 0 && a.addv != nullptr) a.addv[qq] = qq < a.q ? a.b_add[qq] : 0.f;
            if (which == 1 && a.addv != nullptr) a.addv[U_QP + qq] = qq < a.q ? a.q_vec[qq] : 0.f;
            if (which == 2 && a.qv != nullptr) a.qv[qq] = qq < a.q ? a.q_vec[qq] : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
bool user64_supported(int S, int d, int h, int q, const char** why) {
    const int dk = h > 0 ? d / h : 0;
    const char* w = nullptr;
    if (S <= 32 || S > 64) w = "33 <= seq_len <= 64";
    else if (d + 1 > 16 * U_KS || (d & 3)) w = "d_model <= 300, a multiple of 4";
    else if (h < 1 || h * dk != d || dk > 32 || (dk & 1)) w = "even d_k <= 32";
    else if (h > U_CS / 2) w = "n_heads <= 10";
    else if (q > U_QP || (q & 3)) w = "q_dim <= 224, a multiple of 4";
    if (why) *why = w;
    return w == nullptr;
}

size_t user64_qkv_bytes(int n_seq, int h) {
    // Q^T | K^T | V^T fragments [n_seq][h][2][12 KiB], then the context fragments [n_seq][2][U_CS][2 KiB]
    return (size_t)n_seq * h * 2 * 12 * 1024 + (size_t)n_seq * 2 * U_CS * 2048;
}
static size_t user64_ctxf_offset(int n_seq, int h) { return (size_t)n_seq * h * 2 * 12 * 1024; }

size_t user64_fwd_planes_bytes(int h) { return (size_t)(3 * h + U_QT) * U_TILE + 2 * U_QP * 4; }
size_t user64_bwd_planes_bytes(int h) { return (size_t)h * U_BTILE + U_QP * 4; }

int launch_user64_fwd(int n_seq, int S, int d, int h, int q, const float* x, const float* w_qkv, const float* b_qkv,
                      const float* w_add, const float* b_add, const float* q_vec, void* planes, float* ctx, float* t, float* w,
                      void* qkv, float* out, bool train, hipStream_t stream) {
    if (n_seq <= 0) return NRMS_OK;
    char* pb = (char*)planes;
    U64PrepArgs p{};
    p.d = d; p.h = h; p.dk = d / h; p.q = q; p.w_qkv = w_qkv; p.b_qkv = b_qkv; p.w_add = w_add; p.b_add = b_add; p.q_vec = q_vec;
    p.wtiles = (__bf16*)pb; p.addv = (float*)(pb + (size_t)(3 * h + U_QT) * U_TILE);
    {
        TimingScope ts("user64_prep", stream);
        hipLaunchKernelGGL(user64_prep_kernel, dim3(1024), dim3(256), 0, stream, p);
    }
    U64FwdArgs a{};
    a.n_seq = n_seq; a.S = S; a.d = d; a.h = h; a.dk = d / h; a.q = q;
    a.x = x; a.wtiles = pb; a.addv = p.addv; a.ctx = ctx; a.t = t; a.w = w;
    a.qkvf = train ? (b8*)qkv : nullptr;
    a.ctxf = (b8*)((char*)qkv + user64_ctxf_offset(n_seq, h));
    a.out = out;
    a.dbg = 0;
    a.stamps = nullptr;
#ifdef NRMS_U64_EXPERIMENTS
    { const char* e = getenv("NRMS_U64_DBG"); a.dbg = e ? atoi(e) : 0; }
    static unsigned long long* g_stamps = nullptr;
    if (g_stamps == nullptr) (void)hipHostMalloc((void**)&g_stamps, 64 * sizeof(unsigned long long));
    a.stamps = g_stamps;
    if (g_stamps != nullptr && getenv("NRMS_U64_STAMPS") != nullptr) {
        (void)hipStreamSynchronize(stream);
        fprintf(stderr, "[shader clock over the head loop: %.0f MHz]\n", (double)(g_stamps[32 + 2] - g_stamps[32 + 1]) / ((g_stamps[2] - g_stamps[1]) * 0.01));
        fprintf(stderr, "[user64_fwd stamps, us] prologue %.2f | tile0 %.2f | head1 (Q..V barrier %.2f, attention+stores %.2f) | heads total %.2f | zero+reload %.2f | additive %.2f | softmax+pool %.2f\n",
                (g_stamps[1] - g_stamps[0]) * 0.01, 0.0, (g_stamps[8] - g_stamps[6]) * 0.01, (g_stamps[7] - g_stamps[8]) * 0.01,
                (g_stamps[2] - g_stamps[1]) * 0.01, (g_stamps[3] - g_stamps[2]) * 0.01, (g_stamps[4] - g_stamps[3]) * 0.01, (g_stamps[5] - g_stamps[4]) * 0.01);
    }
#endif
    const void* fn = train ? (const void*)user64_fwd_kernel<true> : (const void*)user64_fwd_kernel<false>;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, U_FWD_LDS);
    if (e != hipSuccess) { set_error("user64_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    TimingScope ts("user64_fwd", stream);
    const dim3 grid(cdiv(n_seq, 2));
    if (train) hipLaunchKernelGGL(user64_fwd_kernel<true>, grid, dim3(U_THREADS), U_FWD_LDS, stream, a);
    else hipLaunchKernelGGL(user64_fwd_kernel<false>, grid, dim3(U_THREADS), U_FWD_LDS, stream, a);
    return check_launch("user64_fwd");
}

// dq_partial: cdiv(n_seq, 2) * U_QP floats.  Leaves ds [M] and dqkv [M, 3d]; d(q_vec) is accumulated.
size_t user64_dq_partial_floats(int n_seq) { return (size_t)cdiv(n_seq > 0 ? n_seq : 1, 2) * U_QP; }     // (pitch q <= U_QP)
int launch_user64_bwd(int n_seq, int S, int d, int h, int q, const float* w_add, const float* q_vec, void* planes, const float* dout,
                      const float* t, const float* w, const void* qkv, float* ds, float* dq_partial, float* dq_vec, float* dqkv,
                      hipStream_t stream) {
    if (n_seq <= 0) return NRMS_OK;
    char* pb = (char*)planes;
    U64PrepArgs p{};
    p.d = d; p.h = h; p.dk = d / h; p.q = q; p.w_add = w_add; p.q_vec = q_vec;
    p.btiles = (__bf16*)pb; p.qv = (float*)(pb + (size_t)h * U_BTILE);
    {
        TimingScope ts("user64_prep", stream);
        hipLaunchKernelGGL(user64_prep_kernel, dim3(512), dim3(256), 0, stream, p);
    }
    U64BwdArgs a{};
    a.n_seq = n_seq; a.S = S; a.d = d; a.h = h; a.dk = d / h; a.q = q; a.qscale = 1.0f / sqrtf((float)(d / h));
    a.btiles = pb; a.qv = p.qv; a.dout = dout; a.t = t; a.w = w;
    a.qkvf = (const b8*)qkv; a.ctxf = (const b8*)((const char*)qkv + user64_ctxf_offset(n_seq, h));
    a.ds = ds; a.dq_partial = dq_partial; a.dqkv = dqkv;
    const hipError_t e = hipFuncSetAttribute((const void*)user64_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, U_BWD_LDS);
    if (e != hipSuccess) { set_error("user64_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    const int n_wg = cdiv(n_seq, 2);
    {
        TimingScope ts("user64_bwd", stream);
        hipLaunchKernelGGL(user64_bwd_kernel, dim3(n_wg), dim3(U_THREADS), U_BWD_LDS, stream, a);
    }
    int rc = check_launch("user64_bwd");
    if (rc) return rc;
    return launch_colsum_add(dq_partial, n_wg, q, dq_vec, stream);
}

}  // namespace nrms
