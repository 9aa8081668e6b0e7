// Per-(sequence, head) scaled-dot-product attention, forward and backward, for the tiny
// sequences of NRMS (S = 30 title words / 50 clicked news, d_k = 30).
//
// One WAVE owns one (sequence, head) unit: Q, K, V [S x d_k] are staged into a wave-private
// LDS region (zero-padded to 32-multiples), scores are computed TRANSPOSED
//     S^T[j][i] = sum_d K[j][d] Q[i][d]         (v_mfma_f32_32x32x2_f32)
// so that a query's whole key axis lies in one lane pair (registers x lane^32): the softmax is
// a register reduction plus one cross-half shuffle, and the probability tile is already the
// B operand of the next product  ctx^T[d][i] = sum_j V[j][d] P^T[j][i]  (k-step r of the MFMA
// consumes accumulator register r; the A operand is read from LDS in the matching key order).
// Replaces ScaledDotProductAttention / the head split+concat of MultiHeadSelfAttention
// (/root/reference/MIND_2020/model/nrms_v0.py:13-23,53-58,72-75); no mask, as in v0.
#include "common.h"

namespace nrms {

constexpr int PADSUM_STRIDE = 192;     // >= 3 d_k (d_k <= 64)

struct AttnArgs {
    int n_seq, S, d, h, dk;
    float scale;            // 1/sqrt(d_k)
    const float* qkv;       // [M, 3d] HEAD-MAJOR columns: [head][Q | K | V][d_k]  (see HeadPerm, common.h)
    float* ctx;             // fwd out [M, d]
    Dropout drop;           // fwd: site 1 on ctx
    const float* dctx;      // bwd in  [M, d] gradient w.r.t. the pre-dropout context (mask already applied)
    float* dqkv;            // bwd out [M, 3d], same head-major column order
    // COMPACT backward (NRMS_FLAG_PAD_ROW_ZERO): row m of dQKV goes to row pos[m] of a compact dqkv, rows with
    // pos[m] < 0 (padding tokens) are not written; their column sums -- their whole contribution to the
    // bias gradient -- are accumulated per wave into padsum[wave][PADSUM_STRIDE]
    const int* pos;
    float* padsum;
    // All-padding sequences (NRMS_FLAG_PAD_ROW_ZERO, no attention mask): every Q|K|V row equals the bias, the
    // attention is uniform, ctx = b_v; dS = 0, dQ = dK = 0 and every dV row is the mean of the dO rows.  The
    // forward (ids + bias_hm given) writes dropout(b_v); the COMPACT backward only adds the column sum of dO
    // to the padding-row sums of dV.  With dropout on the probabilities (pdrop) a query's context is
    // b_v x (its kept keys) / (S (1 - p)) and the dO rows enter that sum with the same factors (allpad_keep_factor).
    const int64_t* ids;
    const float* bias_hm;
    int w2, hw;             // float2 per row of one head block (3 d_k / 2) and of one operand (d_k / 2)
    uint32_t magic;         // ceil(2^20 / w2): idx / w2 == (idx * magic) >> 20 for idx < 2^20 / w2
    const uint8_t* mask;    // optional [n_seq, S]: v1's pairwise mask mask_i*mask_j -> masked_fill(-1e9)
                            // (model/nrms_v1.py:27-33); null = v0 (no mask at all)
    Dropout pdrop;          // dropout on the attention PROBABILITIES (nrms_naml.py:36-39), site 2, element
                            // ((seq * h + head) * S + query) * S + key; thresh 0 = none
    bool split;             // products as split-bf16 (the bf16x3 / bf16 modes) instead of exact f32 MFMA
};

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in issue order; this only stops the compiler from
    // moving LDS accesses across the phase boundary.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Register prefetch of one [S x d_k] operand of the NEXT unit (global -> VGPR), written to LDS
// after the current unit's compute: keeps HBM requests in flight while the wave is on the MFMAs.
template <int NS, int ND>
struct Prefetch {
    static constexpr int LPR = 16 * ND, RPI = 64 / LPR, IT = 32 * NS / RPI;
    float2 v[IT];
    // Branch-free: every lane loads from a clamped (always valid) address; the out-of-range lanes are
    // zeroed by a select in store().  A predicated load (`cond ? *p : 0`) makes hipcc branch around
    // EACH load with an exec-mask save/restore and a wait; a select placed HERE makes it wait for all
    // the loads (vmcnt 31..0) before the unit's MFMAs start, i.e. no prefetch at all.  The raw values
    // have no use until store(), so the loads stay in flight across the whole compute phase.
    __device__ __forceinline__ void load(const float* src, long ld, int S, int dk, int lane) {
        const int c2 = lane % LPR, rsub = lane / LPR;
        const float* col = src + (2 * c2 < dk ? 2 * c2 : 0);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int r = rsub + it * RPI;
            v[it] = *reinterpret_cast<const float2*>(col + (long)min(r, S - 1) * ld);
        }
    }
    // unconditional: out-of-range entries are zeros and land on the zero padding (rows < 32 NS,
    // columns < 32 ND are always inside the [SP][RS] image), which they thereby re-establish
    __device__ __forceinline__ void store(float* dst, int RS, int S, int dk, int lane) const {
        const int c2 = lane % LPR, rsub = lane / LPR;
        const bool cok = 2 * c2 < dk;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int r = rsub + it * RPI;
            const bool ok = cok && r < S;
            float2 t;
            t.x = ok ? v[it].x : 0.f;
            t.y = ok ? v[it].y : 0.f;
            *reinterpret_cast<float2*>(dst + r * RS + 2 * c2) = t;
        }
    }
};

// The Q, K and V slices of one head are ONE contiguous [S x 3 d_k] block per row (head-major columns,
// 360 bytes at d_k = 30): lanes walk it densely, 64 consecutive float2 per instruction, so a wave
// instruction touches ~5 cache lines instead of the 8 that four separate 120-byte row slices straddle
// (these kernels are bound by the line-touch rate of the vector memory path, not by bytes).
struct BlockPos { uint32_t goff, inimg, which, row; bool ok; };   // float2 units; which: 0 Q, 1 K, 2 V

template <int NS, int ND>
struct PrefetchQKV {
    static constexpr int IT = 24 * NS * ND;         // >= 32 NS * 48 ND / 64
    // Where the registers allow (32x32 tiles) each iteration's block position is computed ONCE per wave
    // and kept packed in one register: bits 0-14 global offset in float2 (row * ld/2 + c), 15-24 offset
    // inside the operand image in float2 (row * RS/2 + c'), 25-26 operand (0 Q, 1 K, 2 V; 3 = out of
    // range), 27-31 row.  Recomputing it per unit costs ~20 VALU per iteration -- a fifth of the forward.
    static constexpr bool PACKED = NS * ND == 1;
    float2 v[IT];
    uint32_t pos[PACKED ? IT : 1];

    __device__ __forceinline__ static BlockPos locate(int idx, long ld, int RS, int total, int w2, int hw, uint32_t magic) {
        const int ic = min(idx, total - 1);
        const int row = (int)(((uint32_t)ic * magic) >> 20);
        const int c = ic - row * w2;
        const int which = (c >= hw ? 1 : 0) + (c >= 2 * hw ? 1 : 0);
        BlockPos b;
        b.goff = (uint32_t)(row * (int)(ld >> 1) + c);
        b.inimg = (uint32_t)(row * (RS >> 1) + c - which * hw);
        b.which = (uint32_t)which;
        b.row = (uint32_t)row;
        b.ok = idx < total;
        return b;
    }
    __device__ __forceinline__ void init(long ld, int RS, int total, int w2, int hw, uint32_t magic, int lane) {
        if (PACKED) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const BlockPos b = locate(lane + 64 * it, ld, RS, total, w2, hw, magic);
                pos[it] = b.goff | (b.inimg << 15) | ((b.ok ? b.which : 3u) << 25) | (b.row << 27);
            }
        }
    }
    // called at the top of every unit: makes the packed words opaque again, otherwise the compiler hoists
    // their unpacked forms (global offset, LDS address, predicate) out of the unit loop
    __device__ __forceinline__ void touch() {
        if (PACKED) {
#pragma unroll
            for (int it = 0; it < IT; ++it) asm volatile("" : "+v"(pos[it]));
        }
    }
    __device__ __forceinline__ BlockPos at(int it, long ld, int RS, int total, int w2, int hw, uint32_t magic,
                                           int lane) const {
        if (!PACKED) return locate(lane + 64 * it, ld, RS, total, w2, hw, magic);
        const uint32_t p = pos[it];
        BlockPos b;
        b.goff = p & 0x7fffu;
        b.inimg = (p >> 15) & 0x3ffu;
        b.which = (p >> 25) & 3u;
        b.row = p >> 27;
        b.ok = b.which != 3u;
        return b;
    }
    __device__ __forceinline__ void load(const float* src, long ld, int RS, int total, int w2, int hw, uint32_t magic,
                                         int lane) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            // (skipping the iterations past the block -- half of them at S = 40, d_k = 50 -- with a wave-uniform branch was
            // measured: the forward got 13 % slower, every load in a basic block of its own)
            const BlockPos b = at(it, ld, RS, total, w2, hw, magic, lane);    // clamped: raw values only (see Prefetch)
            v[it] = *reinterpret_cast<const float2*>(src + 2 * b.goff);
        }
    }
    // img: the three [SP][RS] images Q, K, V, contiguous
    __device__ __forceinline__ void store(float* img, int SPRS, long ld, int RS, int total, int w2, int hw,
                                          uint32_t magic, int lane) const {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const BlockPos b = at(it, ld, RS, total, w2, hw, magic, lane);
            if (b.ok) *reinterpret_cast<float2*>(img + b.which * SPRS + 2 * b.inimg) = v[it];
        }
    }
};

// Re-establish the zero padding of `n_img` contiguous [SP][RS] images (rows S.., columns d_k..DKP-1):
// the staged outputs and the transpose image of the previous unit have dirtied it.
__device__ __forceinline__ void zero_padding(float* img, int n_img, int SP, int RS, int DKP, int S, int dk, int lane) {
    const float2 z = {0.f, 0.f};
    const int wpad = (DKP - dk) >> 1;              // wave-uniform
    for (int r = lane; r < n_img * SP; r += 64)
        for (int j = 0; j < wpad; ++j) *reinterpret_cast<float2*>(img + r * RS + dk + 2 * j) = z;
    const int per = (SP - S) * RS / 2;
    for (int i = 0; i < n_img; ++i)
        for (int idx = lane; idx < per; idx += 64)
            *reinterpret_cast<float2*>(img + (i * SP + S) * RS + 2 * idx) = z;
}

// ---- split-bf16 products (the bf16x3 / bf16 modes): a 16-deep k-step is three v_mfma_f32_32x32x16_bf16 (hi.hi + hi.lo +
// lo.hi, ~2^-16 relative) instead of eight v_mfma_f32_32x32x2_f32 -- 96 matrix-pipe cycles instead of 512.  A lane supplies
// k = 8 hh + j of the step for both operands; an accumulator tile enters as a B operand through its registers 8s..8s+7
// (k-step s), whose rows are 16 s + 8 (j >> 2) + 4 hh + (j & 3): the A side reads the same rows.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Sp8 { bf16x8 hi, lo; };
__device__ __forceinline__ Sp8 split8(const float (&x)[8]) {
    Sp8 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        s.hi[i] = h;
        s.lo[i] = (__bf16)(x[i] - (float)h);
    }
    return s;
}
__device__ __forceinline__ f32x16 mma3(const Sp8& a, const Sp8& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ int krow16(int s, int j, int hh) { return 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3); }

// S^T (or dP^T) tiles: out[jt][it] += sum_d A[j][d] B[i][d]
template <int NS, int ND, bool SPLIT = false>
__device__ __forceinline__ void abt_tiles(const float* A, const float* B, int RS, int l32, int hh,
                                          f32x16 (&out)[NS][NS]) {
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int it = 0; it < NS; ++it) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (SPLIT) {
#pragma unroll
                for (int t = 0; t < 2 * ND; ++t) {
                    float av[8], bv[8];
                    const float* ap = A + (jt * 32 + l32) * RS + 16 * t + 8 * hh;
                    const float* bp = B + (it * 32 + l32) * RS + 16 * t + 8 * hh;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { av[e] = a0[e]; av[4 + e] = a1[e]; bv[e] = b0[e]; bv[4 + e] = b1[e]; }
                    acc = mma3(split8(av), split8(bv), acc);
                }
                out[jt][it] = acc;
                continue;
            }
#pragma unroll
            for (int t = 0; t < 4 * ND; ++t) {
                // k permutation: lane half hh supplies d = 8t + 4hh + e for MFMA e (both operands)
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + (jt * 32 + l32) * RS + 8 * t + 4 * hh);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(B + (it * 32 + l32) * RS + 8 * t + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(a4[e], b4[e], acc);
            }
            out[jt][it] = acc;
        }
}

// in-register softmax over keys (rows of the transposed tile) for each query column.
// msk (wave-private LDS, 1.0 / 0.0 per position) reproduces v1's masked_fill(mask_i*mask_j == 0, -1e9):
// a fully masked query row ends up uniform over the S real keys, exactly like the reference.
template <int NS, bool MASKED>
__device__ __forceinline__ void softmax_cols(f32x16 (&st)[NS][NS], float scale, int S, int l32, int hh,
                                             const float* msk) {
#pragma unroll
    for (int it = 0; it < NS; ++it) {
        const float mi = MASKED ? msk[it * 32 + l32] : 1.0f;
        float mx = -1e30f;
#pragma unroll
        for (int jt = 0; jt < NS; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + crow32(r, hh);
                float s = st[jt][it][r] * scale;
                if (MASKED) s = (mi * msk[j] != 0.f) ? s : -1e9f;
                s = j < S ? s : -1e30f;
                st[jt][it][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NS; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // exp via v_exp_f32 (exp2): |x| <= ~90 here, relative error ~1e-7 -- far inside the
                // 1e-5 score tolerance, and ~10x fewer VALU instructions than the libm expf
                const float p = __expf(st[jt][it][r] - mx);
                st[jt][it][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int jt = 0; jt < NS; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[jt][it][r] *= inv;
    }
}

// x^T[key j][query i] *= keep(i, j) / (1 - p): the probability-dropout mask of one (sequence, head) unit on a transposed
// tile set.  A lane's registers 4g..4g+3 are four consecutive keys (crow32), i.e. four consecutive elements of the
// flat [.., S, S] index when S is a multiple of 4: one Philox call per register group; otherwise one per element.
template <int NS>
__device__ __forceinline__ void prob_dropout(f32x16 (&xt)[NS][NS], const Dropout& pd, long unit, int S, int l32, int hh) {
#pragma unroll
    for (int it = 0; it < NS; ++it) {
        const uint64_t row0 = (uint64_t)((pd.unit_of(unit) * S + it * 32 + l32) * S);
#pragma unroll
        for (int jt = 0; jt < NS; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int j0 = jt * 32 + 8 * g + 4 * hh;
                const uint64_t e0 = row0 + j0;
                const f32x4 lo = dropout_scale4(pd.seed, 2u, e0 >> 2, pd.thresh, pd.inv_keep);
                if ((S & 3) == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xt[jt][it][4 * g + e] *= lo[e];
                } else {
                    // the four keys straddle two Philox groups: elements sh..3 of this one, 0..sh-1 of the next
                    // (selects, not an indexed array: that would live in scratch)
                    const f32x4 hi = dropout_scale4(pd.seed, 2u, (e0 >> 2) + 1, pd.thresh, pd.inv_keep);
                    const int sh = (int)(e0 & 3);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = e < 4 ? lo[e] : 0.f, a1 = e + 1 < 4 ? lo[e + 1] : hi[e + 1 - 4];
                        const float a2 = e + 2 < 4 ? lo[e + 2] : hi[e + 2 - 4], a3 = e + 3 < 4 ? lo[e + 3] : hi[e + 3 - 4];
                        xt[jt][it][4 * g + e] *= sh == 0 ? a0 : (sh == 1 ? a1 : (sh == 2 ? a2 : a3));
                    }
                }
            }
    }
}

// out^T[dd][i] = sum_j A[j][dd] X^T[j][i], X^T an accumulator tile set (keys in registers)
template <int NS, int ND, bool SPLIT = false>
__device__ __forceinline__ void at_x_tiles(const float* A, int RS, int l32, int hh, const f32x16 (&xt)[NS][NS],
                                           f32x16 (&out)[ND][NS]) {
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int it = 0; it < NS; ++it) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (SPLIT) {
#pragma unroll
                for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float av[8], bv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            av[j] = A[(jt * 32 + krow16(ks, j, hh)) * RS + dt * 32 + l32];
                            bv[j] = xt[jt][it][8 * ks + j];
                        }
                        acc = mma3(split8(av), split8(bv), acc);
                    }
                out[dt][it] = acc;
                continue;
            }
#pragma unroll
            for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float av = A[(jt * 32 + crow32(r, hh)) * RS + dt * 32 + l32];
                    acc = mfma32(av, xt[jt][it][r], acc);
                }
            out[dt][it] = acc;
        }
}

// out^T[dd][j] = sum_i A[i][dd] X[i][j] for an accumulator tile set X^T[j][i] (queries i in lanes): X is
// brought into "queries in rows" order through an LDS image, one 32-column block (jt) at a time -- the image is
// [SP][33] floats whatever the sequence length, so it always fits over the dead V tile (a full [64][65] image
// cost the 64-row kernels a third of their occupancy).
constexpr int TSH = 33;
template <int NS, int ND, bool SPLIT = false>
__device__ __forceinline__ void at_lds_transposed(const float* A, int RS, float* T, const f32x16 (&xt)[NS][NS], int l32,
                                                  int hh, f32x16 (&out)[ND][NS]) {
#pragma unroll
    for (int jt = 0; jt < NS; ++jt) {
        wave_sync();                               // earlier readers of the image (and of the tile it aliases) are done
#pragma unroll
        for (int it = 0; it < NS; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(it * 32 + l32) * TSH + crow32(r, hh)] = xt[jt][it][r];
        wave_sync();
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (SPLIT) {
#pragma unroll
                for (int t = 0; t < 2 * NS; ++t) {
                    float av[8], bv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 16 * t + 8 * hh + j;
                        av[j] = A[i * RS + dt * 32 + l32];
                        bv[j] = T[i * TSH + l32];
                    }
                    acc = mma3(split8(av), split8(bv), acc);
                }
                out[dt][jt] = acc;
                continue;
            }
#pragma unroll 8
            for (int t = 0; t < 16 * NS; ++t) {
                const int i = 2 * t + hh;
                acc = mfma32(A[i * RS + dt * 32 + l32], T[i * TSH + l32], acc);
            }
            out[dt][jt] = acc;
        }
    }
}

// out^T[dd][i] accumulator tiles -> LDS rows [i][dd].  Unconditional: the padding positions get
// finite garbage, which is harmless because the next unit's Prefetch::store rewrites the whole
// [32 NS][32 ND] image (zeros in the padding) before anything reads it, and the global store
// below only reads the valid [S][dk] part.
template <int NS, int ND>
__device__ __forceinline__ void stage_out(float* dst, int RS, const f32x16 (&o)[ND][NS], int S, int dk, int l32,
                                          int hh) {
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int it = 0; it < NS; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(it * 32 + l32) * RS + dt * 32 + crow32(r, hh)] = o[dt][it][r];
}

// ---- query-block forms (the 64-row backward): one 32-query column block `it` of the transposed tile set at a time, so
// that P^T and dS^T of a unit are 2 NS tiles instead of 2 NS^2 -- the full-set form of the 64 x 64 backward needed ~400
// registers for tiles alone and spilled up to 405 of them.  Softmax and D are per query, so a block is self-contained;
// dV and dK (sums over the queries) accumulate across the blocks.
template <int NS, int ND, bool SPLIT>
__device__ __forceinline__ void abt_col(const float* A, const float* Bq, int RS, int l32, int hh, f32x16 (&out)[NS]) {
#pragma unroll
    for (int jt = 0; jt < NS; ++jt) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (SPLIT) {
#pragma unroll
            for (int t = 0; t < 2 * ND; ++t) {
                float av[8], bv[8];
                const float* ap = A + (jt * 32 + l32) * RS + 16 * t + 8 * hh;
                const float* bp = Bq + l32 * RS + 16 * t + 8 * hh;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { av[e] = a0[e]; av[4 + e] = a1[e]; bv[e] = b0[e]; bv[4 + e] = b1[e]; }
                acc = mma3(split8(av), split8(bv), acc);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4 * ND; ++t) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + (jt * 32 + l32) * RS + 8 * t + 4 * hh);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bq + l32 * RS + 8 * t + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(a4[e], b4[e], acc);
            }
        }
        out[jt] = acc;
    }
}

template <int NS, bool MASKED>
__device__ __forceinline__ void softmax_col(f32x16 (&st)[NS], float scale, int S, int it, int l32, int hh, const float* msk) {
    const float mi = MASKED ? msk[it * 32 + l32] : 1.0f;
    float mx = -1e30f;
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jt * 32 + crow32(r, hh);
            float s = st[jt][r] * scale;
            if (MASKED) s = (mi * msk[j] != 0.f) ? s : -1e9f;
            s = j < S ? s : -1e30f;
            st[jt][r] = s;
            mx = fmaxf(mx, s);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(st[jt][r] - mx);
            st[jt][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[jt][r] *= inv;
}

// Probability-dropout decisions of one query block as a bit mask (bit 16 jt + r = keep key crow32(r) of key block jt for this
// lane's query): generated ONCE per block and applied to dP^T and, later, to P^T -- a Philox call is ~100 instructions,
// and the backward needs the same mask twice (one register instead of 16 NS of scales or a second generation).
template <int NS>
__device__ __forceinline__ uint32_t prob_keep_bits_col(const Dropout& pd, long unit, int S, int it, int l32, int hh) {
    static_assert(NS <= 2, "32 mask bits");
    const uint64_t row0 = (uint64_t)((pd.unit_of(unit) * S + it * 32 + l32) * S);
    uint32_t bits = 0;
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint64_t e0 = row0 + jt * 32 + 8 * g + 4 * hh;
            const f32x4 lo = dropout_scale4(pd.seed, 2u, e0 >> 2, pd.thresh, pd.inv_keep);
            if ((S & 3) == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bits |= (lo[e] != 0.f ? 1u : 0u) << (16 * jt + 4 * g + e);
            } else {
                // the four keys straddle two Philox groups: elements sh..3 of this one, 0..sh-1 of the next
                const f32x4 hi = dropout_scale4(pd.seed, 2u, (e0 >> 2) + 1, pd.thresh, pd.inv_keep);
                const int sh = (int)(e0 & 3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = lo[e], a1 = e + 1 < 4 ? lo[(e + 1) & 3] : hi[(e + 1) & 3];
                    const float a2 = e + 2 < 4 ? lo[(e + 2) & 3] : hi[(e + 2) & 3], a3 = e + 3 < 4 ? lo[(e + 3) & 3] : hi[(e + 3) & 3];
                    const float sc = sh == 0 ? a0 : (sh == 1 ? a1 : (sh == 2 ? a2 : a3));
                    bits |= (sc != 0.f ? 1u : 0u) << (16 * jt + 4 * g + e);
                }
            }
        }
    return bits;
}
__device__ __forceinline__ void apply_keep_tile(f32x16& x, uint32_t bits16, float inv_keep) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] *= ((bits16 >> r) & 1u) != 0u ? inv_keep : 0.f;
}

// All-padding unit under probability dropout: every Q|K|V row is the bias, so P = 1/S before the mask and the context of
// query i is b_v x c_i with c_i = (kept keys of query i) / (S (1 - p)); backward: dQ = 0, the dK rows sum to zero (each
// query's dS row sums to zero) and the dV rows sum to sum_i c_i dO_i.  Returns c_i of query 32 it + l32 (both lane halves).
template <int NS>
__device__ __forceinline__ float allpad_keep_factor(const Dropout& pd, long unit, int S, int it, int l32, int hh) {
    const uint32_t bits = prob_keep_bits_col<NS>(pd, unit, S, it, l32, hh);
    int cnt = 0;
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            cnt += (jt * 32 + crow32(r, hh) < S && ((bits >> (16 * jt + r)) & 1u) != 0u) ? 1 : 0;
    cnt += __shfl_xor(cnt, 32, 64);
    return (float)cnt * pd.inv_keep / (float)S;
}
// the factors of all queries of the unit into fac[0..32 NS) (wave-private LDS; 1.0 without probability dropout)
template <int NS>
__device__ __forceinline__ void allpad_factors(const Dropout& pd, long unit, int S, int l32, int hh, float* fac) {
#pragma unroll
    for (int it = 0; it < NS; ++it) {
        const float c = pd.thresh != 0u ? allpad_keep_factor<NS>(pd, unit, S, it, l32, hh) : 1.0f;
        if (hh == 0) fac[it * 32 + l32] = c;
    }
}

// out^T[dd][i] = sum_j A[j][dd] X^T[j][i] for the queries i of one block
template <int NS, int ND, bool SPLIT>
__device__ __forceinline__ void at_x_col(const float* A, int RS, int l32, int hh, const f32x16 (&xt)[NS], f32x16 (&out)[ND]) {
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int jt = 0; jt < NS; ++jt) {
            if (SPLIT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float av[8], bv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        av[j] = A[(jt * 32 + krow16(ks, j, hh)) * RS + dt * 32 + l32];
                        bv[j] = xt[jt][8 * ks + j];
                    }
                    acc = mma3(split8(av), split8(bv), acc);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc = mfma32(A[(jt * 32 + crow32(r, hh)) * RS + dt * 32 + l32], xt[jt][r], acc);
            }
        }
        out[dt] = acc;
    }
}

// out^T[dd][j] += sum over the block's queries i of Aq[i][dd] X[i][j]  (Aq = the block's 32 rows; X^T[j][i] through the
// [32][33] image T, one key block at a time)
template <int NS, int ND, bool SPLIT>
__device__ __forceinline__ void at_lds_transposed_col(const float* Aq, int RS, float* T, const f32x16 (&xt)[NS], int l32, int hh,
                                                      f32x16 (&out)[ND][NS]) {
#pragma unroll
    for (int jt = 0; jt < NS; ++jt) {
        wave_sync();
#pragma unroll
        for (int r = 0; r < 16; ++r) T[l32 * TSH + crow32(r, hh)] = xt[jt][r];
        wave_sync();
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            f32x16 acc = out[dt][jt];
            if (SPLIT) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float av[8], bv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 16 * t + 8 * hh + j;
                        av[j] = Aq[i * RS + dt * 32 + l32];
                        bv[j] = T[i * TSH + l32];
                    }
                    acc = mma3(split8(av), split8(bv), acc);
                }
            } else {
#pragma unroll 8
                for (int t = 0; t < 16; ++t) {
                    const int i = 2 * t + hh;
                    acc = mfma32(Aq[i * RS + dt * 32 + l32], T[i * TSH + l32], acc);
                }
            }
            out[dt][jt] = acc;
        }
    }
}

template <int ND>
__device__ __forceinline__ void stage_out_col(float* dstq, int RS, const f32x16 (&o)[ND], int l32, int hh) {
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dstq[l32 * RS + dt * 32 + crow32(r, hh)] = o[dt][r];
}

// ---------------------------------------------------------------------------------------
template <int NS, int ND, int WPB, bool MASKED, bool APAD, bool SPLIT>
__global__ __launch_bounds__(64 * WPB) void attn_fwd_kernel(AttnArgs a) {
    constexpr int SP = 32 * NS, DKP = 32 * ND, RS = DKP + 4, WF = 3 * SP * RS + 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    float* Qs = lds + wave * WF;
    float* Ks = Qs + SP * RS;
    float* Vs = Ks + SP * RS;
    float* Ms = Vs + SP * RS;
    for (int i = lane; i < WF; i += 64) Qs[i] = 0.f;
    wave_sync();

    const long total = (long)a.n_seq * a.h;
    const long ld = 3L * a.d;
    const long ustride = (long)gridDim.x * WPB;
    PrefetchQKV<NS, ND> pf;
    const int blk = a.S * a.w2;                 // float2 in one head block
    long u = (long)blockIdx.x * WPB + wave;
    // APAD: all-padding units neither load nor compute.  `ap_cur` / `ap_next` = this / the next unit is all
    // padding; idv = this lane's token id two units ahead (in flight for a whole unit before its ballot).
    bool ap_cur = false, ap_next = false;
    long idv = 1;
    auto seq_of = [&](long uu) { return min(uu, total - 1) / a.h; };
    if (u < total) {
        const long seq = u / a.h;
        if (APAD) {
            const long id0 = lane < a.S ? a.ids[seq * a.S + lane] : 0;
            const long id1 = lane < a.S ? a.ids[seq_of(u + ustride) * a.S + lane] : 0;
            idv = lane < a.S ? a.ids[seq_of(u + 2 * ustride) * a.S + lane] : 0;
            ap_cur = __ballot(id0 != 0) == 0ull;
            ap_next = __ballot(id1 != 0) == 0ull;
        }
        pf.init(ld, RS, blk, a.w2, a.hw, a.magic, lane);
        if (!ap_cur)
            pf.load(a.qkv + seq * a.S * ld + (int)(u - seq * a.h) * 3 * a.dk, ld, RS, blk, a.w2, a.hw, a.magic, lane);
    }
    for (; u < total; u += ustride) {
        const long seq = u / a.h;
        const int head = (int)(u - seq * a.h);
        // opaque copy of the lane id: where the positions are not kept packed (big tiles) they are
        // recomputed for every unit instead of being hoisted out of the loop into ~100 live registers
        int lv = lane;
        asm volatile("" : "+v"(lv));
        pf.touch();
        const bool allpad = ap_cur;
        if (!allpad) {
            pf.store(Qs, SP * RS, ld, RS, blk, a.w2, a.hw, a.magic, lv);
            zero_padding(Qs, 3, SP, RS, DKP, a.S, a.dk, lane);
            if (MASKED) Ms[lane] = (lane < a.S && a.mask[seq * a.S + lane] != 0) ? 1.0f : 0.0f;
        }
        wave_sync();
        {                                        // next unit's operands fly while this one computes
            const long un = min(u + ustride, total - 1), sn = un / a.h;
            if (!ap_next)
                pf.load(a.qkv + sn * a.S * ld + (int)(un - sn * a.h) * 3 * a.dk, ld, RS, blk, a.w2, a.hw, a.magic, lv);
            if (APAD) {
                ap_cur = ap_next;
                ap_next = __ballot(idv != 0) == 0ull;            // ids of unit u + 2 ustride, loaded a unit ago
                idv = lane < a.S ? a.ids[seq_of(u + 3 * ustride) * a.S + lane] : 0;
            }
        }

        if (!allpad) {
            f32x16 st[NS][NS];
            abt_tiles<NS, ND, SPLIT>(Ks, Qs, RS, l32, hh, st);
            softmax_cols<NS, MASKED>(st, a.scale, a.S, l32, hh, Ms);
            if (a.pdrop.thresh != 0u) prob_dropout<NS>(st, a.pdrop, u, a.S, l32, hh);
            f32x16 o[ND][NS];
            at_x_tiles<NS, ND, SPLIT>(Vs, RS, l32, hh, st, o);
            wave_sync();
            stage_out<NS, ND>(Qs, RS, o, a.S, a.dk, l32, hh);
        } else {
            // uniform attention over identical rows: every context row is the V bias of this head (times the query's
            // kept-key factor under probability dropout; Ms is free: the shortcut needs an unmasked unit)
            allpad_factors<NS>(a.pdrop, u, a.S, l32, hh, Ms);
            wave_sync();
            const float* bv = a.bias_hm + head * 3 * a.dk + 2 * a.dk;
            for (int i = lane; i < a.S * a.hw; i += 64) {
                const int r = i / a.hw, c2 = i - r * a.hw;
                float2 v = *reinterpret_cast<const float2*>(bv + 2 * c2);
                v.x *= Ms[r]; v.y *= Ms[r];
                *reinterpret_cast<float2*>(Qs + r * RS + 2 * c2) = v;
            }
        }
        wave_sync();

        // store of ctx[seq*S + r][head*dk + c] through dropout site 1
        if (ND == 1) {
            // One lane per Philox GROUP (4 consecutive elements of the flat [M, d] index, 16-byte aligned in the
            // ctx row): 8 groups cover a head's <= 32 columns, 8 rows per wave-instruction.  A lane makes ONE
            // Philox call for its four elements and stores them as one float4 (the head's first / last group may
            // be half outside the head: then a float2) -- half the Philox calls and half the store instructions
            // of a float2-per-lane loop.
            const int gi = lane & 7, rsub = lane >> 3;
            const int hb = head * a.dk;
            const int o0 = hb & 3;                                   // 0 or 2 (d_k is even, d % 4 == 0)
            const int clo = 4 * gi - o0, chi = clo + 2;              // head-relative columns of the two float2 halves
            const bool vlo = clo >= 0 && clo < a.dk, vhi = chi < a.dk;
            for (int r = rsub; r < a.S; r += 8) {
                const long m = seq * a.S + r;
                float2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
                if (vlo) lo = *reinterpret_cast<const float2*>(Qs + r * RS + clo);
                if (vhi) hi = *reinterpret_cast<const float2*>(Qs + r * RS + chi);
                if (a.drop.thresh != 0u) {
                    uint32_t rnd[4];
                    philox4x32_7(a.drop.seed, (uint64_t)(m * a.d + hb + clo) >> 2, 1u, rnd);
                    lo.x = rnd[0] >= a.drop.thresh ? lo.x * a.drop.inv_keep : 0.f;
                    lo.y = rnd[1] >= a.drop.thresh ? lo.y * a.drop.inv_keep : 0.f;
                    hi.x = rnd[2] >= a.drop.thresh ? hi.x * a.drop.inv_keep : 0.f;
                    hi.y = rnd[3] >= a.drop.thresh ? hi.y * a.drop.inv_keep : 0.f;
                }
                float* o = a.ctx + m * a.d + hb + clo;
                if (vlo && vhi) *reinterpret_cast<f32x4*>(o) = f32x4{lo.x, lo.y, hi.x, hi.y};
                else if (vlo) *reinterpret_cast<float2*>(o) = lo;
                else if (vhi) *reinterpret_cast<float2*>(o + 2) = hi;
            }
        } else {
            constexpr int LPR = 16 * ND, RPI = 64 / LPR;
            const int c2 = lane % LPR, rsub = lane / LPR;
            if (2 * c2 < a.dk) {
                for (int r = rsub; r < a.S; r += RPI) {
                    float2 v = *reinterpret_cast<const float2*>(Qs + r * RS + 2 * c2);
                    const long m = seq * a.S + r;
                    const int col = head * a.dk + 2 * c2;
                    if (a.drop.thresh != 0u) {
                        uint32_t rnd[4];
                        const uint64_t e = (uint64_t)(m * a.d + col);
                        philox4x32_7(a.drop.seed, e >> 2, 1u, rnd);
                        const int q = (int)(e & 3);          // 0 or 2: col is even
                        v.x = (q == 0 ? rnd[0] : rnd[2]) >= a.drop.thresh ? v.x * a.drop.inv_keep : 0.f;
                        v.y = (q == 0 ? rnd[1] : rnd[3]) >= a.drop.thresh ? v.y * a.drop.inv_keep : 0.f;
                    }
                    *reinterpret_cast<float2*>(a.ctx + m * a.d + col) = v;
                }
            }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------
// Backward: recompute P from Q,K (no S x S tensor is ever stored), then
//   dP^T = V dO^T ; D_i = sum_j P_ij dP_ij ; dS = P (dP - D) / sqrt(d_k)
//   dV^T = dO^T P ; dQ^T = K^T dS^T ; dK^T = Q^T dS
// The two products that sum over the query index need P / dS with queries in rows: one
// 32x32 transpose through a wave-private LDS image each.
// (second launch bound: the 32x32 instantiations must stay at two waves per SIMD, i.e. <= 256 registers; the
// COMPACT variants would otherwise settle at 260 and lose half their occupancy)
template <int NS, int ND, int WPB, bool MASKED, bool COMPACT, bool SPLIT>
__global__ __launch_bounds__(64 * WPB, NS * ND == 1 ? 2 : 1) void attn_bwd_kernel(AttnArgs a) {
    constexpr bool PF = NS == 1;      // prefetch across the compute only where registers allow
    constexpr int SP = 32 * NS, DKP = 32 * ND, RS = DKP + 4, TS = TSH;
    // the 32x32 transpose image fits inside the V region once V is dead (after dP^T): no LDS of its
    // own -> 18.7 KB per wave instead of 22.9 KB, i.e. 8 waves per CU instead of 6 (this kernel is a
    // long chain of LDS round trips, so it lives off occupancy)
    // (the 64-row form walks the queries in blocks and needs V in every block: its [32][33] image has a region of its own)
    constexpr bool ALIAS = NS == 1;
    constexpr int WF = 4 * SP * RS + (ALIAS ? 0 : 32 * TS) + 64 + (COMPACT ? 64 + 3 * DKP : 0);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    float* Qs = lds + wave * WF;
    float* Ks = Qs + SP * RS;
    float* Vs = Ks + SP * RS;
    float* Gs = Vs + SP * RS;
    float* Tb = ALIAS ? Vs : Gs + SP * RS;
    float* Ms = Gs + SP * RS + (ALIAS ? 0 : 32 * TS);
    float* dQs = NS == 1 ? Ks : Qs;                     // where dQ is staged (see the two forms below)
    // 64-column heads (ROWIO): the Q|K|V block of a row is walked by (row, lane) instead of a flat index -- a lane's column,
    // operand and LDS offsets are constants of the wave, a row's compact position is wave-uniform.  (The flat walk of
    // PrefetchQKV recomputes row / operand / offsets per element: 96 iterations of ~45 instructions and two dependent LDS
    // round trips each per unit, with one wave per SIMD to hide them -- most of this kernel's time at S = 40, d_k = 50.)
    constexpr bool ROWIO = ND == 2;
    constexpr bool TOPLOAD = ROWIO && !PF;
    constexpr int KR = 2;                               // float2 per row: 3 d_k / 2 <= 96
    int rcol[KR], rin[KR], rout[KR];
    if (ROWIO) {
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int c = lane + 64 * k, cc = min(c, a.w2 - 1);
            const int which = (cc >= a.hw ? 1 : 0) + (cc >= 2 * a.hw ? 1 : 0), col = cc - which * a.hw;
            rcol[k] = c < a.w2 ? 2 * cc : -1;
            rin[k] = which * SP * RS + 2 * col;
            rout[k] = (which == 0 ? (int)(dQs - Qs) : (which == 1 ? (int)(Gs - Qs) : (int)(Vs - Qs))) + 2 * col;
        }
    }
    int* Ps = reinterpret_cast<int*>(Ms + 64);          // COMPACT: compact row of each row of the unit (-1 = padding)
    float* Acc = reinterpret_cast<float*>(Ps + 64);    // COMPACT: [3][DKP] column sums of the padding rows of dQ, dK, dV
                                                        // (kept in LDS: three more registers would cost a wave per SIMD)
    for (int i = lane; i < WF; i += 64) Qs[i] = 0.f;
    wave_sync();

    const long total = (long)a.n_seq * a.h;
    const long ld = 3L * a.d;
    const long ustride = (long)gridDim.x * WPB;
    PrefetchQKV<NS, ND> pf;
    Prefetch<NS, ND> pg;
    const int blk = a.S * a.w2;
    long u = (long)blockIdx.x * WPB + wave;
    // COMPACT: compact-row numbers (pos) of this lane's row in the current / next / next-but-one unit, so
    // that neither the unit's own map nor the "is the next unit all padding?" test waits for memory
    int p_cur = 0, p_n1 = 0, p_n2 = 0;
    auto pos_of = [&](long uu) { return lane < a.S ? a.pos[(min(uu, total - 1) / a.h) * a.S + lane] : -1; };
    bool ap_next = false;
    if (u < total) {
        const long seq = u / a.h;
        const int head = (int)(u - seq * a.h);
        bool ap0 = false;
        if (COMPACT) {
            p_cur = pos_of(u); p_n1 = pos_of(u + ustride); p_n2 = pos_of(u + 2 * ustride);
            ap0 = !MASKED && __ballot(p_cur >= 0) == 0ull;
        }
        pf.init(ld, RS, blk, a.w2, a.hw, a.magic, lane);
        if (!TOPLOAD) {
            if (!ap0) pf.load(a.qkv + seq * a.S * ld + head * 3 * a.dk, ld, RS, blk, a.w2, a.hw, a.magic, lane);
            pg.load(a.dctx + seq * a.S * a.d + head * a.dk, a.d, a.S, a.dk, lane);
        }
    }
    for (; u < total; u += ustride) {
        const long seq = u / a.h;
        const int head = (int)(u - seq * a.h);
        int lv = lane;                           // opaque: see attn_fwd_kernel
        asm volatile("" : "+v"(lv));
        pf.touch();
        bool allpad = false;
        if (COMPACT) {
            Ps[lane] = lane < a.S ? p_cur : 0;
            allpad = !MASKED && __ballot(p_cur >= 0) == 0ull;        // no live token in this sequence
            ap_next = !MASKED && __ballot(p_n1 >= 0) == 0ull;
        }
        // 64 x 64 tiles: no registers to carry the next unit across the compute (a conditional load at the end of the
        // previous iteration kept all 192 of them live through it) -- this unit's operands are loaded here
        if (TOPLOAD) pg.load(a.dctx + seq * a.S * a.d + head * a.dk, a.d, a.S, a.dk, lane);
        if (!allpad) {
            if (TOPLOAD) {
                const float* src = a.qkv + seq * a.S * ld + head * 3 * a.dk;
                constexpr int RB = 32;                          // 64 row pieces in flight: S = 40 pays the memory latency twice
                for (int r0 = 0; r0 < a.S; r0 += RB) {
                    float2 t[RB][KR];
#pragma unroll
                    for (int i = 0; i < RB; ++i)
#pragma unroll
                        for (int k = 0; k < KR; ++k)
                            t[i][k] = *reinterpret_cast<const float2*>(src + (long)min(r0 + i, a.S - 1) * ld + max(rcol[k], 0));
#pragma unroll
                    for (int i = 0; i < RB; ++i)
#pragma unroll
                        for (int k = 0; k < KR; ++k)
                            if (rcol[k] >= 0 && r0 + i < a.S) *reinterpret_cast<float2*>(Qs + rin[k] + (r0 + i) * RS) = t[i][k];
                }
            } else {
                pf.store(Qs, SP * RS, ld, RS, blk, a.w2, a.hw, a.magic, lv);
            }
            zero_padding(Qs, 3, SP, RS, DKP, a.S, a.dk, lane);
        }
        pg.store(Gs, RS, a.S, a.dk, lane);      // dctx arrives already masked (dctx GEMM epilogue)
        if (MASKED) Ms[lane] = (lane < a.S && a.mask[seq * a.S + lane] != 0) ? 1.0f : 0.0f;
        wave_sync();
        if (PF) {                                // (an all-padding next unit only needs its dO rows)
            const long un = min(u + ustride, total - 1), sn = un / a.h;
            const int hn = (int)(un - sn * a.h);
            if (!ap_next) pf.load(a.qkv + sn * a.S * ld + hn * 3 * a.dk, ld, RS, blk, a.w2, a.hw, a.magic, lv);
            pg.load(a.dctx + sn * a.S * a.d + hn * a.dk, a.d, a.S, a.dk, lane);
        }
        if (COMPACT) { p_cur = p_n1; p_n1 = p_n2; p_n2 = pos_of(u + 3 * ustride); }

        if (!allpad) {
        const float* msk = Ms;
        if constexpr (NS > 1) {
            // query-block form: per block of 32 queries P^T, dP^T -> dS^T, the block's dQ (complete), and its
            // contribution to dV and dK; dQ is staged over the block's own Q rows once dK has read them
            f32x16 dv[ND][NS], dkk[ND][NS];
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { dv[dt][jt][r] = 0.f; dkk[dt][jt][r] = 0.f; }
#pragma unroll
            for (int it = 0; it < NS; ++it) {
                f32x16 st[NS], dp[NS];
                abt_col<NS, ND, SPLIT>(Ks, Qs + it * 32 * RS, RS, l32, hh, st);
                softmax_col<NS, MASKED>(st, a.scale, a.S, it, l32, hh, msk);          // st = P^T
                abt_col<NS, ND, SPLIT>(Vs, Gs + it * 32 * RS, RS, l32, hh, dp);        // dp = dP^T (of the dropped P)
                uint32_t keep = 0;
                if (a.pdrop.thresh != 0u) {
                    keep = prob_keep_bits_col<NS>(a.pdrop, u, a.S, it, l32, hh);
#pragma unroll
                    for (int jt = 0; jt < NS; ++jt) apply_keep_tile(dp[jt], keep >> (16 * jt), a.pdrop.inv_keep);
                }
                float D = 0.f;
#pragma unroll
                for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) D += st[jt][r] * dp[jt][r];
                D += __shfl_xor(D, 32, 64);
#pragma unroll
                for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float dsv = st[jt][r] * (dp[jt][r] - D) * a.scale;                                     // dS^T
                        if (MASKED && msk[it * 32 + l32] * msk[jt * 32 + crow32(r, hh)] == 0.f) dsv = 0.f;
                        dp[jt][r] = dsv;
                    }
                if (a.pdrop.thresh != 0u) {                                         // dV wants the dropped P
#pragma unroll
                    for (int jt = 0; jt < NS; ++jt) apply_keep_tile(st[jt], keep >> (16 * jt), a.pdrop.inv_keep);
                }
                at_lds_transposed_col<NS, ND, SPLIT>(Gs + it * 32 * RS, RS, Tb, st, l32, hh, dv);       // dV^T += dO^T P
                at_lds_transposed_col<NS, ND, SPLIT>(Qs + it * 32 * RS, RS, Tb, dp, l32, hh, dkk);      // dK^T += Q^T dS
                f32x16 dq[ND];
                at_x_col<NS, ND, SPLIT>(Ks, RS, l32, hh, dp, dq);                                       // dQ^T = K^T dS^T
                wave_sync();
                stage_out_col<ND>(Qs + it * 32 * RS, RS, dq, l32, hh);
            }
            wave_sync();                                           // K, V, dO are dead
            stage_out<NS, ND>(Gs, RS, dkk, a.S, a.dk, l32, hh);
            stage_out<NS, ND>(Vs, RS, dv, a.S, a.dk, l32, hh);
            wave_sync();
        } else {
        f32x16 st[NS][NS], dp[NS][NS];
        abt_tiles<NS, ND, SPLIT>(Ks, Qs, RS, l32, hh, st);
        softmax_cols<NS, MASKED>(st, a.scale, a.S, l32, hh, msk);  // st = P^T
        abt_tiles<NS, ND, SPLIT>(Vs, Gs, RS, l32, hh, dp);         // dp = dP^T
        // probability dropout: the product above is the gradient of the DROPPED probabilities; through the mask it is
        // dP.  The keep decisions are generated once, as 16 bits, and used again below (for dV)
        uint32_t keep = 0;
        if (a.pdrop.thresh != 0u) {
            keep = prob_keep_bits_col<1>(a.pdrop, u, a.S, 0, l32, hh);
            apply_keep_tile(dp[0][0], keep, a.pdrop.inv_keep);
        }
#pragma unroll
        for (int it = 0; it < NS; ++it) {
            float D = 0.f;
#pragma unroll
            for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) D += st[jt][it][r] * dp[jt][it][r];
            D += __shfl_xor(D, 32, 64);
#pragma unroll
            for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float dsv = st[jt][it][r] * (dp[jt][it][r] - D) * a.scale;                           // dS^T
                    // masked_fill passes no gradient to the scores it overwrote
                    if (MASKED && msk[it * 32 + l32] * msk[jt * 32 + crow32(r, hh)] == 0.f) dsv = 0.f;
                    dp[jt][it][r] = dsv;
                }
        }
        // dV^T = dO^T P  (queries summed: P through the transpose image, which may live in the dead V
        // region); dV stays in registers until that image has also served dK
        f32x16 dv[ND][NS];
        if (a.pdrop.thresh != 0u) apply_keep_tile(st[0][0], keep, a.pdrop.inv_keep);   // P is dead after dS: dV wants dropped P
        at_lds_transposed<NS, ND, SPLIT>(Gs, RS, Tb, st, l32, hh, dv);
        // dQ^T = K^T dS^T  (keys summed: dS^T straight from registers); K is dead afterwards -> stage dQ there
        {
            f32x16 dq[ND][NS];
            at_x_tiles<NS, ND, SPLIT>(Ks, RS, l32, hh, dp, dq);
            wave_sync();
            stage_out<NS, ND>(Ks, RS, dq, a.S, a.dk, l32, hh);
        }
        // dK^T = Q^T dS ; dO is dead -> stage dK there
        {
            f32x16 dkk[ND][NS];
            at_lds_transposed<NS, ND, SPLIT>(Qs, RS, Tb, dp, l32, hh, dkk);
            wave_sync();
            stage_out<NS, ND>(Gs, RS, dkk, a.S, a.dk, l32, hh);
        }
        stage_out<NS, ND>(Vs, RS, dv, a.S, a.dk, l32, hh);     // transpose image is dead now
        wave_sync();
        }

        // dQ | dK | dV of this head: one contiguous block per row again, written densely
        {
            float* ob = a.dqkv + (COMPACT ? 0 : seq * a.S * ld) + head * 3 * a.dk;
            constexpr int OIT = ROWIO ? 0 : 24 * NS * ND;
            if (ROWIO) {
                for (int r0 = 0; r0 < a.S; r0 += 4) {
                    int crow[4];
                    float2 t[4][KR];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // row r of the unit lives in row Ps[r] of the compact dqkv (-1: a padding token, no row)
                        const int r = min(r0 + i, a.S - 1);
                        crow[i] = COMPACT ? __builtin_amdgcn_readfirstlane(Ps[r]) : r;
                        if (r0 + i >= a.S) crow[i] = -1;
#pragma unroll
                        for (int k = 0; k < KR; ++k) t[i][k] = *reinterpret_cast<const float2*>(Qs + rout[k] + r * RS);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (crow[i] >= 0) {
#pragma unroll
                            for (int k = 0; k < KR; ++k)
                                if (rcol[k] >= 0) *reinterpret_cast<float2*>(ob + (long)crow[i] * ld + rcol[k]) = t[i][k];
                        }
                }
            }
#pragma unroll                                   // full: a run-time `it` would push pos[] into scratch
            for (int it = 0; it < OIT; ++it) {
                // (the scheduler would otherwise hoist all 24 strip reads ahead of the stores: +8 live registers,
                // which is one wave per SIMD at this kernel's 256)
                if (COMPACT && (it & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                const BlockPos b = pf.at(it, ld, RS, blk, a.w2, a.hw, a.magic, lv);
                const float* img = b.which == 0 ? dQs : (b.which == 1 ? Gs : Vs);   // dQ, dK, dV staging images
                if (COMPACT) {
                    // same block, but row r of the unit lives in row Ps[r] of the compact dqkv
                    const int crow = Ps[b.ok ? b.row : 0];
                    if (b.ok && crow >= 0)
                        *reinterpret_cast<float2*>(ob + 2 * ((long)b.goff + ((long)crow - (long)b.row) * (ld >> 1))) =
                            *reinterpret_cast<const float2*>(img + 2 * b.inimg);
                } else if (b.ok) {
                    *reinterpret_cast<float2*>(ob + 2 * b.goff) = *reinterpret_cast<const float2*>(img + 2 * b.inimg);
                }
            }
            if (COMPACT) {
                // padding rows: accumulate their dQ | dK | dV column sums (this wave always works on the same
                // head: the launcher makes the unit stride a multiple of h).  Lane = one column of the staged
                // images (and, for 32-column tiles, one half of the rows): independent LDS reads, no chain.
                constexpr int RG = ND == 1 ? 2 : 1;
                const int col = ND == 1 ? l32 : lane, rg = ND == 1 ? hh : 0;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
                for (int r = rg; r < a.S; r += RG) {
                    const float m = Ps[r] < 0 ? 1.0f : 0.0f;
                    s0 += m * dQs[r * RS + col];
                    s1 += m * Gs[r * RS + col];
                    s2 += m * Vs[r * RS + col];
                }
                if (ND == 1) {
                    s0 += __shfl_xor(s0, 32, 64);
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                }
                if (ND != 1 || hh == 0) {
                    Acc[col] += s0;
                    Acc[DKP + col] += s1;
                    Acc[2 * DKP + col] += s2;
                }
            }
        }
        } else {
            // all-padding sequence: identical Q|K|V rows => uniform attention, dS = 0, dQ = dK = 0 and every
            // dV row = mean of the dO rows; nothing is written (padding rows), the rows' dV sum = sum of dO
            // (under probability dropout the rows enter with their kept-key factors; Ms is free: unmasked unit)
            allpad_factors<NS>(a.pdrop, u, a.S, l32, hh, Ms);
            wave_sync();
            constexpr int RG = ND == 1 ? 2 : 1;
            const int col = ND == 1 ? l32 : lane, rg = ND == 1 ? hh : 0;
            float s2 = 0.f;
#pragma unroll 4
            for (int r = rg; r < a.S; r += RG) s2 += Ms[r] * Gs[r * RS + col];
            if (ND == 1) s2 += __shfl_xor(s2, 32, 64);
            if (ND != 1 || hh == 0) Acc[2 * DKP + col] += s2;
        }
        wave_sync();
        if (!PF && !TOPLOAD) {                // 64 x 32 tiles: no registers to spare, load after the compute
            const long un = min(u + ustride, total - 1), sn = un / a.h;
            const int hn = (int)(un - sn * a.h);
            if (!ap_next) pf.load(a.qkv + sn * a.S * ld + hn * 3 * a.dk, ld, RS, blk, a.w2, a.hw, a.magic, lv);
            pg.load(a.dctx + sn * a.S * a.d + hn * a.dk, a.d, a.S, a.dk, lane);
        }
    }
    if (COMPACT) {
        float* out = a.padsum + ((long)blockIdx.x * WPB + wave) * PADSUM_STRIDE;
        const int col = ND == 1 ? l32 : lane;
        if (col < a.dk && (ND != 1 || hh == 0)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) out[i * a.dk + col] = Acc[i * DKP + col];
        }
    }
}

// ---------------------------------------------------------------------------------------
// dbias[perm.src(n')] += sum over the waves of head(n') of their padding-row column sums (fixed order)
__global__ __launch_bounds__(256) void padsum_reduce_kernel(const float* padsum, int n_waves, int h, int dk, HeadPerm perm,
                                                            float* dbias) {
    // one wave per output column: lanes stride over the waves of that head, fixed-order tree at the end
    const int lane = threadIdx.x & 63;
    const int np = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (np >= 3 * h * dk) return;
    const int head = np / (3 * dk), cc = np - head * 3 * dk;
    float s = 0.f;
    for (int w = head + lane * h; w < n_waves; w += 64 * h) s += padsum[(long)w * PADSUM_STRIDE + cc];
    s = wave_sum(s);
    if (lane == 0) dbias[perm.src(np)] += s;
}

// ---------------------------------------------------------------------------------------
// Backward of the 64 x 64 units (S > 32 and d_k > 32: nrms_naml's abstracts, S = 40, d_k = 50), TWO waves per unit.
// A wave's four [64][68] operand images are 70 KB of LDS: with a wave per unit only two units fit a CU and two of its four
// SIMDs sit idle, each wave a single dependent chain of LDS round trips and MFMAs.  Here wave w owns query block w of the
// unit (the block form above): its P^T / dS^T, the block's dQ (complete) and its share of dV and dK (sums over ITS
// queries); the two shares meet in the staging images (wave 0 stores dK and wave 1 dV, then each adds its share of the
// other, a fixed order).  Same LDS per unit, every SIMD busy, half the time per unit.
template <bool MASKED, bool COMPACT, bool SPLIT>
__global__ __launch_bounds__(128, 1) void attn_bwd_coop_kernel(AttnArgs a) {
    constexpr int NS = 2, ND = 2, SP = 64, DKP = 64, RS = DKP + 4, TS = TSH, KR = 2;
    constexpr int WF = 4 * SP * RS + 2 * 32 * TS + 64 + (COMPACT ? 64 + 3 * DKP : 0);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    float* Qs = lds;
    float* Ks = Qs + SP * RS;
    float* Vs = Ks + SP * RS;
    float* Gs = Vs + SP * RS;
    float* Tb = Gs + SP * RS + w * 32 * TS;              // this wave's transpose image
    float* Ms = Gs + SP * RS + 2 * 32 * TS;
    int* Ps = reinterpret_cast<int*>(Ms + 64);
    float* Acc = reinterpret_cast<float*>(Ps + 64);
    for (int i = threadIdx.x; i < WF; i += 128) Qs[i] = 0.f;
    __syncthreads();

    const long total = (long)a.n_seq * a.h;
    const long ld = 3L * a.d;
    int rcol[KR], rin[KR], rout[KR];                    // see ROWIO in attn_bwd_kernel; dQ is staged over Q
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int c = lane + 64 * k, cc = min(c, a.w2 - 1);
        const int which = (cc >= a.hw ? 1 : 0) + (cc >= 2 * a.hw ? 1 : 0), col = cc - which * a.hw;
        rcol[k] = c < a.w2 ? 2 * cc : -1;
        rin[k] = which * SP * RS + 2 * col;
        rout[k] = (which == 0 ? 0 : (which == 1 ? 3 * SP * RS : 2 * SP * RS)) + 2 * col;     // dQ in Qs, dK in Gs, dV in Vs
    }
    const int Sw = min(32, a.S - 32 * w);               // rows of this wave's query block (S > 32: at least 1)
    float* Qw = Qs + w * 32 * RS;
    float* Gw = Gs + w * 32 * RS;
    auto pos_of = [&](long uu) { return lane < a.S ? a.pos[(min(uu, total - 1) / a.h) * a.S + lane] : -1; };
    int p_cur = 0, p_n1 = 0;
    if (COMPACT) { p_cur = pos_of(blockIdx.x); p_n1 = pos_of((long)blockIdx.x + gridDim.x); }

    for (long u = blockIdx.x; u < total; u += gridDim.x) {
        const long seq = u / a.h;
        const int head = (int)(u - seq * a.h);
        bool allpad = false;
        if (COMPACT) {
            if (w == 0) Ps[lane] = lane < a.S ? p_cur : 0;
            allpad = !MASKED && __ballot(p_cur >= 0) == 0ull;        // no live token in this sequence
        }
        // each wave brings its block's dO rows and its half of the Q|K|V rows
        Prefetch<1, ND> pg;
        pg.load(a.dctx + (seq * a.S + 32 * w) * a.d + head * a.dk, a.d, Sw, a.dk, lane);
        if (!allpad) {
            const float* src = a.qkv + seq * a.S * ld + head * 3 * a.dk;
            const int r0 = 32 * w;
            float2 t[32][KR];
#pragma unroll
            for (int i = 0; i < 32; ++i)
#pragma unroll
                for (int k = 0; k < KR; ++k)
                    t[i][k] = *reinterpret_cast<const float2*>(src + (long)min(r0 + i, a.S - 1) * ld + max(rcol[k], 0));
#pragma unroll
            for (int i = 0; i < 32; ++i)
#pragma unroll
                for (int k = 0; k < KR; ++k)
                    if (rcol[k] >= 0 && r0 + i < a.S) *reinterpret_cast<float2*>(Qs + rin[k] + (r0 + i) * RS) = t[i][k];
            if (w == 0) zero_padding(Qs, 2, SP, RS, DKP, a.S, a.dk, lane);
            else zero_padding(Vs, 1, SP, RS, DKP, a.S, a.dk, lane);
        }
        pg.store(Gw, RS, Sw, a.dk, lane);               // dctx arrives already masked (dctx GEMM epilogue)
        if (MASKED && w == 0) Ms[lane] = (lane < a.S && a.mask[seq * a.S + lane] != 0) ? 1.0f : 0.0f;
        if (COMPACT) { p_cur = p_n1; p_n1 = pos_of(u + 2L * gridDim.x); }
        __syncthreads();

        if (!allpad) {
            const float* msk = Ms;
            f32x16 st[NS], dp[NS];
            abt_col<NS, ND, SPLIT>(Ks, Qw, RS, l32, hh, st);
            softmax_col<NS, MASKED>(st, a.scale, a.S, w, l32, hh, msk);              // st = P^T
            abt_col<NS, ND, SPLIT>(Vs, Gw, RS, l32, hh, dp);                          // dp = dP^T (of the dropped P)
            uint32_t keep = 0;
            if (a.pdrop.thresh != 0u) {
                keep = prob_keep_bits_col<NS>(a.pdrop, u, a.S, w, l32, hh);
#pragma unroll
                for (int jt = 0; jt < NS; ++jt) apply_keep_tile(dp[jt], keep >> (16 * jt), a.pdrop.inv_keep);
            }
            float D = 0.f;
#pragma unroll
            for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) D += st[jt][r] * dp[jt][r];
            D += __shfl_xor(D, 32, 64);
#pragma unroll
            for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float dsv = st[jt][r] * (dp[jt][r] - D) * a.scale;                                         // dS^T
                    if (MASKED && msk[w * 32 + l32] * msk[jt * 32 + crow32(r, hh)] == 0.f) dsv = 0.f;
                    dp[jt][r] = dsv;
                }
            if (a.pdrop.thresh != 0u) {                                             // dV wants the dropped P
#pragma unroll
                for (int jt = 0; jt < NS; ++jt) apply_keep_tile(st[jt], keep >> (16 * jt), a.pdrop.inv_keep);
            }
            f32x16 dv[ND][NS], dkk[ND][NS];
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int jt = 0; jt < NS; ++jt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { dv[dt][jt][r] = 0.f; dkk[dt][jt][r] = 0.f; }
            at_lds_transposed_col<NS, ND, SPLIT>(Gw, RS, Tb, st, l32, hh, dv);           // this block's dO^T P
            at_lds_transposed_col<NS, ND, SPLIT>(Qw, RS, Tb, dp, l32, hh, dkk);          // this block's Q^T dS
            f32x16 dq[ND];
            at_x_col<NS, ND, SPLIT>(Ks, RS, l32, hh, dp, dq);                            // dQ^T = K^T dS^T, complete
            wave_sync();
            stage_out_col<ND>(Qw, RS, dq, l32, hh);
            __syncthreads();                                   // K, V, dO are dead in both waves
            if (w == 0) stage_out<NS, ND>(Gs, RS, dkk, a.S, a.dk, l32, hh);
            else stage_out<NS, ND>(Vs, RS, dv, a.S, a.dk, l32, hh);
            __syncthreads();
            {
                float* dst = w == 0 ? Vs : Gs;                 // the other wave's share is there: add this one's
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                    for (int it = 0; it < NS; ++it)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dst[(it * 32 + l32) * RS + dt * 32 + crow32(r, hh)] += w == 0 ? dv[dt][it][r] : dkk[dt][it][r];
            }
            __syncthreads();

            // dQ | dK | dV of this head, one contiguous block per row; the waves take alternate groups of four rows
            float* ob = a.dqkv + (COMPACT ? 0 : seq * a.S * ld) + head * 3 * a.dk;
            for (int r0 = 4 * w; r0 < a.S; r0 += 8) {
                int crow[4];
                float2 t[4][KR];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = min(r0 + i, a.S - 1);
                    crow[i] = COMPACT ? __builtin_amdgcn_readfirstlane(Ps[r]) : r;
                    if (r0 + i >= a.S) crow[i] = -1;
#pragma unroll
                    for (int k = 0; k < KR; ++k) t[i][k] = *reinterpret_cast<const float2*>(Qs + rout[k] + r * RS);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (crow[i] >= 0) {
#pragma unroll
                        for (int k = 0; k < KR; ++k)
                            if (rcol[k] >= 0) *reinterpret_cast<float2*>(ob + (long)crow[i] * ld + rcol[k]) = t[i][k];
                    }
            }
            if (COMPACT) {
                // padding rows: their dQ | dK (wave 0) and dV (wave 1) column sums, one column per lane
                float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
                for (int r = 0; r < a.S; ++r) {
                    const float m = Ps[r] < 0 ? 1.0f : 0.0f;
                    if (w == 0) { s0 += m * Qs[r * RS + lane]; s1 += m * Gs[r * RS + lane]; }
                    else s0 += m * Vs[r * RS + lane];
                }
                if (w == 0) { Acc[lane] += s0; Acc[DKP + lane] += s1; }
                else Acc[2 * DKP + lane] += s0;
            }
        } else {
            // all-padding sequence: dQ = 0, the dK rows sum to zero, the dV rows sum to sum_i c_i dO_i (c_i = 1 without
            // probability dropout, see allpad_keep_factor); wave w supplies the factors of its query block
            const float c = a.pdrop.thresh != 0u ? allpad_keep_factor<NS>(a.pdrop, u, a.S, w, l32, hh) : 1.0f;
            if (hh == 0) Ms[w * 32 + l32] = c;
            __syncthreads();
            if (w == 1) {
                float s2 = 0.f;
#pragma unroll 4
                for (int r = 0; r < a.S; ++r) s2 += Ms[r] * Gs[r * RS + lane];
                Acc[2 * DKP + lane] += s2;
            }
        }
        __syncthreads();
    }
    if (COMPACT && w == 0 && lane < a.dk) {
        float* out = a.padsum + (long)blockIdx.x * PADSUM_STRIDE;
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i * a.dk + lane] = Acc[i * DKP + lane];
    }
}

template <bool MASKED, bool COMPACT>
static int launch_attn_bwd_coop2(const AttnArgs& a, float* dbias, hipStream_t stream) {
    constexpr size_t bytes = (4 * 64 * 68 + 2 * 32 * 33 + 64 + (COMPACT ? 64 + 3 * 64 : 0)) * sizeof(float);
    const long total = (long)a.n_seq * a.h;
    int blocks = (int)(total < 256 * 16 ? total : 256 * 16);
    if (COMPACT)                            // every workgroup keeps one head: unit stride = multiple of h
        while (blocks % a.h != 0) ++blocks;
    auto kern = a.split ? attn_bwd_coop_kernel<MASKED, COMPACT, true> : attn_bwd_coop_kernel<MASKED, COMPACT, false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    {
        TimingScope ts("attn_bwd", stream);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), bytes, stream, a);
        const int rc = check_launch("attn_bwd");
        if (rc) return rc;
    }
    if (COMPACT) {
        TimingScope ts("padsum_reduce", stream);
        hipLaunchKernelGGL(padsum_reduce_kernel, dim3(cdiv(3 * a.d, 4)), dim3(256), 0, stream, a.padsum, blocks, a.h, a.dk,
                           HeadPerm{a.dk, a.h}, dbias);
        return check_launch("padsum_reduce");
    }
    return NRMS_OK;
}

static int launch_attn_bwd_coop(const AttnArgs& a, float* dbias, hipStream_t stream) {
    if (a.pos != nullptr)
        return a.mask != nullptr ? launch_attn_bwd_coop2<true, true>(a, dbias, stream) : launch_attn_bwd_coop2<false, true>(a, dbias, stream);
    return a.mask != nullptr ? launch_attn_bwd_coop2<true, false>(a, dbias, stream) : launch_attn_bwd_coop2<false, false>(a, dbias, stream);
}

// ---------------------------------------------------------------------------------------
// Forward of the 64 x 64 units, two waves per unit (see attn_bwd_coop_kernel): wave w loads half of the Q|K|V rows and
// computes the context rows of query block w -- scores against all keys, softmax, probability dropout, P V -- which
// it stages over its own Q rows and writes out.  No operand prefetch across units (it cost 192 registers and ~45
// instructions per float2 in the flat walk); six waves per CU instead of three hide the load instead.
template <bool MASKED, bool APAD, bool SPLIT>
__global__ __launch_bounds__(128, 2) void attn_fwd_coop_kernel(AttnArgs a) {
    constexpr int NS = 2, ND = 2, SP = 64, DKP = 64, RS = DKP + 4, KR = 2;
    constexpr int WF = 3 * SP * RS + 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    float* Qs = lds;
    float* Ks = Qs + SP * RS;
    float* Vs = Ks + SP * RS;
    float* Ms = Vs + SP * RS;
    for (int i = threadIdx.x; i < WF; i += 128) Qs[i] = 0.f;
    __syncthreads();

    const long total = (long)a.n_seq * a.h;
    const long ld = 3L * a.d;
    int rcol[KR], rin[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int c = lane + 64 * k, cc = min(c, a.w2 - 1);
        const int which = (cc >= a.hw ? 1 : 0) + (cc >= 2 * a.hw ? 1 : 0), col = cc - which * a.hw;
        rcol[k] = c < a.w2 ? 2 * cc : -1;
        rin[k] = which * SP * RS + 2 * col;
    }
    const int rend = min(a.S, 32 * w + 32);             // this wave's rows: [32 w, rend)
    float* Qw = Qs + w * 32 * RS;
    auto id_of = [&](long uu) { return lane < a.S ? a.ids[(min(uu, total - 1) / a.h) * a.S + lane] : 0; };
    long id_cur = 1, id_n1 = 1;
    if (APAD) { id_cur = id_of(blockIdx.x); id_n1 = id_of((long)blockIdx.x + gridDim.x); }

    for (long u = blockIdx.x; u < total; u += gridDim.x) {
        const long seq = u / a.h;
        const int head = (int)(u - seq * a.h);
        const bool allpad = APAD && __ballot(id_cur != 0) == 0ull;
        if (APAD) { id_cur = id_n1; id_n1 = id_of(u + 2L * gridDim.x); }
        if (!allpad) {
            const float* src = a.qkv + seq * a.S * ld + head * 3 * a.dk;
            const int r0 = 32 * w;
            float2 t[32][KR];
#pragma unroll
            for (int i = 0; i < 32; ++i)
#pragma unroll
                for (int k = 0; k < KR; ++k)
                    t[i][k] = *reinterpret_cast<const float2*>(src + (long)min(r0 + i, a.S - 1) * ld + max(rcol[k], 0));
#pragma unroll
            for (int i = 0; i < 32; ++i)
#pragma unroll
                for (int k = 0; k < KR; ++k)
                    if (rcol[k] >= 0 && r0 + i < a.S) *reinterpret_cast<float2*>(Qs + rin[k] + (r0 + i) * RS) = t[i][k];
            if (w == 0) zero_padding(Qs, 2, SP, RS, DKP, a.S, a.dk, lane);
            else zero_padding(Vs, 1, SP, RS, DKP, a.S, a.dk, lane);
            if (MASKED && w == 0) Ms[lane] = (lane < a.S && a.mask[seq * a.S + lane] != 0) ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (!allpad) {
            f32x16 st[NS];
            abt_col<NS, ND, SPLIT>(Ks, Qw, RS, l32, hh, st);
            softmax_col<NS, MASKED>(st, a.scale, a.S, w, l32, hh, Ms);
            if (a.pdrop.thresh != 0u) {
                const uint32_t keep = prob_keep_bits_col<NS>(a.pdrop, u, a.S, w, l32, hh);
#pragma unroll
                for (int jt = 0; jt < NS; ++jt) apply_keep_tile(st[jt], keep >> (16 * jt), a.pdrop.inv_keep);
            }
            f32x16 o[ND];
            at_x_col<NS, ND, SPLIT>(Vs, RS, l32, hh, st, o);
            wave_sync();
            stage_out_col<ND>(Qw, RS, o, l32, hh);
        } else {
            // uniform attention over identical rows: every context row is the V bias of this head (times the query's
            // kept-key factor under probability dropout); this wave's rows, factors in its half of Ms
            const float c = a.pdrop.thresh != 0u ? allpad_keep_factor<NS>(a.pdrop, u, a.S, w, l32, hh) : 1.0f;
            if (hh == 0) Ms[w * 32 + l32] = c;
            wave_sync();
            const float* bv = a.bias_hm + head * 3 * a.dk + 2 * a.dk;
            for (int i = lane; i < 32 * a.hw; i += 64) {
                const int r = i / a.hw, c2 = i - r * a.hw;
                float2 v = *reinterpret_cast<const float2*>(bv + 2 * c2);
                v.x *= Ms[w * 32 + r]; v.y *= Ms[w * 32 + r];
                *reinterpret_cast<float2*>(Qw + r * RS + 2 * c2) = v;
            }
        }
        wave_sync();
        // ctx[seq*S + r][head*dk + c] of this wave's rows through dropout site 1: 32 lanes per row
        {
            const int c2 = l32;
            if (2 * c2 < a.dk) {
                for (int r = 32 * w + hh; r < rend; r += 2) {
                    float2 v = *reinterpret_cast<const float2*>(Qs + r * RS + 2 * c2);
                    const long m = seq * a.S + r;
                    const int col = head * a.dk + 2 * c2;
                    if (a.drop.thresh != 0u) {
                        uint32_t rnd[4];
                        const uint64_t e = (uint64_t)(m * a.d + col);
                        philox4x32_7(a.drop.seed, e >> 2, 1u, rnd);
                        const int q = (int)(e & 3);          // 0 or 2: col is even
                        v.x = (q == 0 ? rnd[0] : rnd[2]) >= a.drop.thresh ? v.x * a.drop.inv_keep : 0.f;
                        v.y = (q == 0 ? rnd[1] : rnd[3]) >= a.drop.thresh ? v.y * a.drop.inv_keep : 0.f;
                    }
                    *reinterpret_cast<float2*>(a.ctx + m * a.d + col) = v;
                }
            }
        }
        __syncthreads();                                   // the other wave may still be reading K and V
    }
}

template <bool MASKED, bool APAD>
static int launch_attn_fwd_coop2(const AttnArgs& a, hipStream_t stream) {
    constexpr size_t bytes = (3 * 64 * 68 + 64) * sizeof(float);
    const long total = (long)a.n_seq * a.h;
    const int blocks = (int)(total < 256 * 24 ? total : 256 * 24);
    auto kern = a.split ? attn_fwd_coop_kernel<MASKED, APAD, true> : attn_fwd_coop_kernel<MASKED, APAD, false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("attn_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    TimingScope ts("attn_fwd", stream);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), bytes, stream, a);
    return check_launch("attn_fwd");
}

static int launch_attn_fwd_coop(const AttnArgs& a, hipStream_t stream) {
    if (a.mask != nullptr) return launch_attn_fwd_coop2<true, false>(a, stream);     // (the shortcut needs an unmasked unit)
    return a.ids != nullptr ? launch_attn_fwd_coop2<false, true>(a, stream) : launch_attn_fwd_coop2<false, false>(a, stream);
}

size_t attention_padsum_floats() { return (size_t)(256 * 16 + 256) * 4 * PADSUM_STRIDE; }   // grid cap + up to n_heads - 1 extra blocks, 4 waves each

template <int NS, int ND, int WPB, bool BWD, bool MASKED, bool COMPACT, bool SPLIT>
static int launch_attn_inst4(const AttnArgs& a, float* dbias, hipStream_t stream) {
    constexpr int SP = 32 * NS, RS = 32 * ND + 4;
    constexpr bool ALIAS = NS == 1;                     // 32-row units: the transpose image lives over the dead V tile
    constexpr size_t wf = BWD ? (4 * SP * RS + (ALIAS ? 0 : 32 * 33) + 64 + (COMPACT ? 64 + 3 * 32 * ND : 0)) : (3 * SP * RS + 64);
    constexpr size_t bytes = wf * WPB * sizeof(float);
    const long total = (long)a.n_seq * a.h;
    int blocks = (int)((total + WPB - 1) / WPB);
    const int cap = 256 * 16;               // persistent-ish: waves walk units with a grid stride
    if (blocks > cap) blocks = cap;
    if (COMPACT && BWD)                     // every wave keeps one head: unit stride = multiple of h
        while ((blocks * WPB) % a.h != 0) ++blocks;
    const char* name = BWD ? "attn_bwd" : "attn_fwd";
    hipError_t e;
    if (BWD) e = hipFuncSetAttribute((const void*)attn_bwd_kernel<NS, ND, WPB, MASKED, COMPACT, SPLIT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    else e = hipFuncSetAttribute((const void*)attn_fwd_kernel<NS, ND, WPB, MASKED, COMPACT && !MASKED, SPLIT>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return NRMS_ELAUNCH; }
    {
        TimingScope ts(name, stream);
        if (BWD) hipLaunchKernelGGL((attn_bwd_kernel<NS, ND, WPB, MASKED, COMPACT, SPLIT>), dim3(blocks), dim3(64 * WPB), bytes, stream, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<NS, ND, WPB, MASKED, COMPACT && !MASKED, SPLIT>), dim3(blocks), dim3(64 * WPB), bytes, stream, a);
        const int rc = check_launch(name);
        if (rc) return rc;
    }
    if (COMPACT && BWD) {
        TimingScope ts("padsum_reduce", stream);
        hipLaunchKernelGGL(padsum_reduce_kernel, dim3(cdiv(3 * a.d, 4)), dim3(256), 0, stream, a.padsum, blocks * WPB, a.h,
                           a.dk, HeadPerm{a.dk, a.h}, dbias);
        return check_launch("padsum_reduce");
    }
    return NRMS_OK;
}

template <int NS, int ND, int WPB, bool BWD, bool MASKED, bool COMPACT>
static int launch_attn_inst3(const AttnArgs& a, float* dbias, hipStream_t stream) {
    return a.split ? launch_attn_inst4<NS, ND, WPB, BWD, MASKED, COMPACT, true>(a, dbias, stream)
                   : launch_attn_inst4<NS, ND, WPB, BWD, MASKED, COMPACT, false>(a, dbias, stream);
}

template <int NS, int ND, int WPB, bool BWD>
static int launch_attn_inst(const AttnArgs& a, float* dbias, hipStream_t stream) {
    if (BWD ? a.pos != nullptr : a.ids != nullptr)     // compact backward / all-padding shortcut in the forward
        return a.mask != nullptr ? launch_attn_inst3<NS, ND, WPB, BWD, true, true>(a, dbias, stream)
                                 : launch_attn_inst3<NS, ND, WPB, BWD, false, true>(a, dbias, stream);
    return a.mask != nullptr ? launch_attn_inst3<NS, ND, WPB, BWD, true, false>(a, dbias, stream)
                             : launch_attn_inst3<NS, ND, WPB, BWD, false, false>(a, dbias, stream);
}

// forward: ids + bias_hm (both or none) enable the all-padding-sequence shortcut;
// backward: pos / padsum / dbias (all three or none) = compact dQKV, see AttnArgs
int launch_attention(bool bwd, int n_seq, int S, int d, int h, const float* qkv, float* ctx, const Dropout& drop,
                     const float* dctx, float* dqkv, const uint8_t* mask, const int64_t* ids, const float* bias_hm,
                     const int* pos, float* padsum, float* dbias, hipStream_t stream, const Dropout* pdrop, bool split) {
    AttnArgs a;
    a.split = split;
    a.mask = mask;
    a.pdrop = pdrop != nullptr ? *pdrop : make_dropout(0, 0.f);
    a.n_seq = n_seq; a.S = S; a.d = d; a.h = h; a.dk = d / h;
    a.scale = 1.0f / sqrtf((float)a.dk);
    a.qkv = qkv; a.ctx = ctx; a.drop = drop; a.dctx = dctx; a.dqkv = dqkv;
    a.pos = pos; a.padsum = padsum; a.ids = ids; a.bias_hm = bias_hm;
    a.hw = a.dk / 2; a.w2 = 3 * a.hw;
    a.magic = a.w2 > 0 ? (uint32_t)(((1u << 20) + a.w2 - 1) / a.w2) : 0;
    if (n_seq <= 0) return NRMS_OK;
    if (S < 1 || S > 64 || a.dk > 64 || (a.dk & 1)) {
        set_error("attention: unsupported S=%d d_k=%d (need 1<=S<=64, even d_k<=64)", S, a.dk);
        return NRMS_EINVAL;
    }
    const int ns = S <= 32 ? 1 : 2, nd = a.dk <= 32 ? 1 : 2;
    if (!bwd) {
        if (ns == 1 && nd == 1) return launch_attn_inst<1, 1, 4, false>(a, dbias, stream);
        if (ns == 2 && nd == 1) return launch_attn_inst<2, 1, 2, false>(a, dbias, stream);
        if (ns == 1 && nd == 2) return launch_attn_inst<1, 2, 2, false>(a, dbias, stream);
        return launch_attn_fwd_coop(a, stream);          // 64 x 64 units: two waves per unit
    }
    if (ns == 1 && nd == 1) return launch_attn_inst<1, 1, 2, true>(a, dbias, stream);
    if (ns == 2 && nd == 1) return launch_attn_inst<2, 1, 1, true>(a, dbias, stream);
    if (ns == 1 && nd == 2) return launch_attn_inst<1, 2, 1, true>(a, dbias, stream);
    return launch_attn_bwd_coop(a, dbias, stream);      // 64 x 64 units: two waves per unit
}

}  // namespace nrms
