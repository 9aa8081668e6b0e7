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

struct AttnArgs {
    int n_seq, S, d, h, dk;
    float scale;            // 1/sqrt(d_k)
    const float* qkv;       // [M, 3d]
    float* ctx;             // fwd out [M, d]
    Dropout drop;           // fwd: site 1 on ctx
    const float* dctx;      // bwd in  [M, d] gradient w.r.t. the pre-dropout context (mask already applied)
    float* dqkv;            // bwd out [M, 3d]
    const uint8_t* mask;    // optional [n_seq, S]: v1's pairwise mask mask_i*mask_j -> masked_fill(-1e9)
                            // (model/nrms_v1.py:27-33); null = v0 (no mask at all)
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

// S^T (or dP^T) tiles: out[jt][it] += sum_d A[j][d] B[i][d]
template <int NS, int ND>
__device__ __forceinline__ void abt_tiles(const float* A, const float* B, int RS, int l32, int hh,
                                          f32x16 (&out)[NS][NS]) {
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int it = 0; it < NS; ++it) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
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

// out^T[dd][i] = sum_j A[j][dd] X^T[j][i], X^T an accumulator tile set (keys in registers)
template <int NS, int ND>
__device__ __forceinline__ void at_x_tiles(const float* A, int RS, int l32, int hh, const f32x16 (&xt)[NS][NS],
                                           f32x16 (&out)[ND][NS]) {
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int it = 0; it < NS; ++it) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
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

// out^T[dd][j] = sum_i A[i][dd] T[i][j], T a row-major [SP][SP+1] LDS image
template <int NS, int ND>
__device__ __forceinline__ void at_lds_tiles(const float* A, int RS, const float* T, int l32, int hh,
                                             f32x16 (&out)[ND][NS]) {
    constexpr int TS = 32 * NS + 1;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int jt = 0; jt < NS; ++jt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
            for (int t = 0; t < 16 * NS; ++t) {
                const int i = 2 * t + hh;
                acc = mfma32(A[i * RS + dt * 32 + l32], T[i * TS + jt * 32 + l32], acc);
            }
            out[dt][jt] = acc;
        }
}

// accumulator tile set X^T[j][i] -> LDS image T[i][j]
template <int NS>
__device__ __forceinline__ void transpose_to_lds(float* T, const f32x16 (&xt)[NS][NS], int l32, int hh) {
    constexpr int TS = 32 * NS + 1;
#pragma unroll
    for (int jt = 0; jt < NS; ++jt)
#pragma unroll
        for (int it = 0; it < NS; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(it * 32 + l32) * TS + jt * 32 + crow32(r, hh)] = xt[jt][it][r];
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

// ---------------------------------------------------------------------------------------
template <int NS, int ND, int WPB, bool MASKED>
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
    Prefetch<NS, ND> pq, pk, pv;
    long u = (long)blockIdx.x * WPB + wave;
    if (u < total) {
        const long seq = u / a.h;
        const float* base = a.qkv + seq * a.S * ld + (int)(u - seq * a.h) * a.dk;
        pq.load(base, ld, a.S, a.dk, lane);
        pk.load(base + a.d, ld, a.S, a.dk, lane);
        pv.load(base + 2 * a.d, ld, a.S, a.dk, lane);
    }
    for (; u < total; u += ustride) {
        const long seq = u / a.h;
        const int head = (int)(u - seq * a.h);
        pq.store(Qs, RS, a.S, a.dk, lane);
        pk.store(Ks, RS, a.S, a.dk, lane);
        pv.store(Vs, RS, a.S, a.dk, lane);
        if (MASKED) Ms[lane] = (lane < a.S && a.mask[seq * a.S + lane] != 0) ? 1.0f : 0.0f;
        wave_sync();
        if (u + ustride < total) {            // next unit's operands fly while this one computes
            const long un = u + ustride, sn = un / a.h;
            const float* base = a.qkv + sn * a.S * ld + (int)(un - sn * a.h) * a.dk;
            pq.load(base, ld, a.S, a.dk, lane);
            pk.load(base + a.d, ld, a.S, a.dk, lane);
            pv.load(base + 2 * a.d, ld, a.S, a.dk, lane);
        }

        f32x16 st[NS][NS];
        abt_tiles<NS, ND>(Ks, Qs, RS, l32, hh, st);
        softmax_cols<NS, MASKED>(st, a.scale, a.S, l32, hh, Ms);
        f32x16 o[ND][NS];
        at_x_tiles<NS, ND>(Vs, RS, l32, hh, st, o);
        wave_sync();
        stage_out<NS, ND>(Qs, RS, o, a.S, a.dk, l32, hh);
        wave_sync();

        // coalesced-as-possible store of ctx[seq*S + r][head*dk + c], dropout site 1
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
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------
// Backward: recompute P from Q,K (no S x S tensor is ever stored), then
//   dP^T = V dO^T ; D_i = sum_j P_ij dP_ij ; dS = P (dP - D) / sqrt(d_k)
//   dV^T = dO^T P ; dQ^T = K^T dS^T ; dK^T = Q^T dS
// The two products that sum over the query index need P / dS with queries in rows: one
// 32x32 transpose through a wave-private LDS image each.
template <int NS, int ND, int WPB, bool MASKED>
__global__ __launch_bounds__(64 * WPB) void attn_bwd_kernel(AttnArgs a) {
    constexpr bool PF = NS == 1;      // prefetch across the compute only where registers allow
    constexpr int SP = 32 * NS, DKP = 32 * ND, RS = DKP + 4, TS = SP + 1;
    // the 32x32 transpose image fits inside the V region once V is dead (after dP^T): no LDS of its
    // own -> 18.7 KB per wave instead of 22.9 KB, i.e. 8 waves per CU instead of 6 (this kernel is a
    // long chain of LDS round trips, so it lives off occupancy)
    constexpr bool ALIAS = SP * TS <= SP * RS;
    constexpr int WF = 4 * SP * RS + (ALIAS ? 0 : SP * TS) + 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l32 = lane & 31, hh = lane >> 5;
    float* Qs = lds + wave * WF;
    float* Ks = Qs + SP * RS;
    float* Vs = Ks + SP * RS;
    float* Gs = Vs + SP * RS;
    float* Tb = ALIAS ? Vs : Gs + SP * RS;
    float* Ms = Gs + SP * RS + (ALIAS ? 0 : SP * TS);
    for (int i = lane; i < WF; i += 64) Qs[i] = 0.f;
    wave_sync();

    const long total = (long)a.n_seq * a.h;
    const long ld = 3L * a.d;
    const long ustride = (long)gridDim.x * WPB;
    Prefetch<NS, ND> pq, pk, pv, pg;
    long u = (long)blockIdx.x * WPB + wave;
    if (u < total) {
        const long seq = u / a.h;
        const int head = (int)(u - seq * a.h);
        const float* base = a.qkv + seq * a.S * ld + head * a.dk;
        pq.load(base, ld, a.S, a.dk, lane);
        pk.load(base + a.d, ld, a.S, a.dk, lane);
        pv.load(base + 2 * a.d, ld, a.S, a.dk, lane);
        pg.load(a.dctx + seq * a.S * a.d + head * a.dk, a.d, a.S, a.dk, lane);
    }
    for (; u < total; u += ustride) {
        const long seq = u / a.h;
        const int head = (int)(u - seq * a.h);
        pq.store(Qs, RS, a.S, a.dk, lane);
        pk.store(Ks, RS, a.S, a.dk, lane);
        pv.store(Vs, RS, a.S, a.dk, lane);
        pg.store(Gs, RS, a.S, a.dk, lane);      // dctx arrives already masked (dctx GEMM epilogue)
        if (MASKED) Ms[lane] = (lane < a.S && a.mask[seq * a.S + lane] != 0) ? 1.0f : 0.0f;
        wave_sync();
        if (PF && u + ustride < total) {
            const long un = u + ustride, sn = un / a.h;
            const int hn = (int)(un - sn * a.h);
            const float* base = a.qkv + sn * a.S * ld + hn * a.dk;
            pq.load(base, ld, a.S, a.dk, lane);
            pk.load(base + a.d, ld, a.S, a.dk, lane);
            pv.load(base + 2 * a.d, ld, a.S, a.dk, lane);
            pg.load(a.dctx + sn * a.S * a.d + hn * a.dk, a.d, a.S, a.dk, lane);
        }

        f32x16 st[NS][NS], dp[NS][NS];
        abt_tiles<NS, ND>(Ks, Qs, RS, l32, hh, st);
        const float* msk = Ms;
        softmax_cols<NS, MASKED>(st, a.scale, a.S, l32, hh, msk);  // st = P^T
        abt_tiles<NS, ND>(Vs, Gs, RS, l32, hh, dp);         // dp = dP^T
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
        wave_sync();
        transpose_to_lds<NS>(Tb, st, l32, hh);
        wave_sync();
        f32x16 dv[ND][NS];
        at_lds_tiles<NS, ND>(Gs, RS, Tb, l32, hh, dv);
        // dQ^T = K^T dS^T  (keys summed: dS^T straight from registers); K is dead afterwards -> stage dQ there
        {
            f32x16 dq[ND][NS];
            at_x_tiles<NS, ND>(Ks, RS, l32, hh, dp, dq);
            wave_sync();
            stage_out<NS, ND>(Ks, RS, dq, a.S, a.dk, l32, hh);
        }
        // dK^T = Q^T dS ; dO is dead -> stage dK there
        transpose_to_lds<NS>(Tb, dp, l32, hh);
        wave_sync();
        {
            f32x16 dkk[ND][NS];
            at_lds_tiles<NS, ND>(Qs, RS, Tb, l32, hh, dkk);
            wave_sync();
            stage_out<NS, ND>(Gs, RS, dkk, a.S, a.dk, l32, hh);
        }
        stage_out<NS, ND>(Vs, RS, dv, a.S, a.dk, l32, hh);     // transpose image is dead now
        wave_sync();

        constexpr int LPR = 16 * ND, RPI = 64 / LPR;
        const int c2 = lane % LPR, rsub = lane / LPR;
        if (2 * c2 < a.dk) {
            float* ob = a.dqkv + seq * a.S * ld + head * a.dk + 2 * c2;
            for (int r = rsub; r < a.S; r += RPI) {
                *reinterpret_cast<float2*>(ob + r * ld) = *reinterpret_cast<const float2*>(Ks + r * RS + 2 * c2);
                *reinterpret_cast<float2*>(ob + r * ld + a.d) = *reinterpret_cast<const float2*>(Gs + r * RS + 2 * c2);
                *reinterpret_cast<float2*>(ob + r * ld + 2 * a.d) = *reinterpret_cast<const float2*>(Vs + r * RS + 2 * c2);
            }
        }
        wave_sync();
        if (!PF && u + ustride < total) {     // big tiles: no registers to spare, load after the compute
            const long un = u + ustride, sn = un / a.h;
            const int hn = (int)(un - sn * a.h);
            const float* base = a.qkv + sn * a.S * ld + hn * a.dk;
            pq.load(base, ld, a.S, a.dk, lane);
            pk.load(base + a.d, ld, a.S, a.dk, lane);
            pv.load(base + 2 * a.d, ld, a.S, a.dk, lane);
            pg.load(a.dctx + sn * a.S * a.d + hn * a.dk, a.d, a.S, a.dk, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------
template <int NS, int ND, int WPB, bool BWD, bool MASKED>
static int launch_attn_inst2(const AttnArgs& a, hipStream_t stream) {
    constexpr int SP = 32 * NS, RS = 32 * ND + 4;
    constexpr bool ALIAS = SP * (SP + 1) <= SP * RS;
    constexpr size_t wf = BWD ? (4 * SP * RS + (ALIAS ? 0 : SP * (SP + 1)) + 64) : (3 * SP * RS + 64);
    constexpr size_t bytes = wf * WPB * sizeof(float);
    const long total = (long)a.n_seq * a.h;
    int blocks = (int)((total + WPB - 1) / WPB);
    const int cap = 256 * 16;               // persistent-ish: waves walk units with a grid stride
    if (blocks > cap) blocks = cap;
    const char* name = BWD ? "attn_bwd" : "attn_fwd";
    hipError_t e;
    if (BWD) e = hipFuncSetAttribute((const void*)attn_bwd_kernel<NS, ND, WPB, MASKED>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    else e = hipFuncSetAttribute((const void*)attn_fwd_kernel<NS, ND, WPB, MASKED>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return NRMS_ELAUNCH; }
    TimingScope ts(name, stream);
    if (BWD) hipLaunchKernelGGL((attn_bwd_kernel<NS, ND, WPB, MASKED>), dim3(blocks), dim3(64 * WPB), bytes, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<NS, ND, WPB, MASKED>), dim3(blocks), dim3(64 * WPB), bytes, stream, a);
    return check_launch(name);
}

template <int NS, int ND, int WPB, bool BWD>
static int launch_attn_inst(const AttnArgs& a, hipStream_t stream) {
    return a.mask != nullptr ? launch_attn_inst2<NS, ND, WPB, BWD, true>(a, stream)
                             : launch_attn_inst2<NS, ND, WPB, BWD, false>(a, stream);
}

int launch_attention(bool bwd, int n_seq, int S, int d, int h, const float* qkv, float* ctx, const Dropout& drop,
                     const float* dctx, float* dqkv, const uint8_t* mask, hipStream_t stream) {
    AttnArgs a;
    a.mask = mask;
    a.n_seq = n_seq; a.S = S; a.d = d; a.h = h; a.dk = d / h;
    a.scale = 1.0f / sqrtf((float)a.dk);
    a.qkv = qkv; a.ctx = ctx; a.drop = drop; a.dctx = dctx; a.dqkv = dqkv;
    if (n_seq <= 0) return NRMS_OK;
    if (S < 1 || S > 64 || a.dk > 64 || (a.dk & 1)) {
        set_error("attention: unsupported S=%d d_k=%d (need 1<=S<=64, even d_k<=64)", S, a.dk);
        return NRMS_EINVAL;
    }
    const int ns = S <= 32 ? 1 : 2, nd = a.dk <= 32 ? 1 : 2;
    if (!bwd) {
        if (ns == 1 && nd == 1) return launch_attn_inst<1, 1, 4, false>(a, stream);
        if (ns == 2 && nd == 1) return launch_attn_inst<2, 1, 2, false>(a, stream);
        if (ns == 1 && nd == 2) return launch_attn_inst<1, 2, 2, false>(a, stream);
        return launch_attn_inst<2, 2, 1, false>(a, stream);
    }
    if (ns == 1 && nd == 1) return launch_attn_inst<1, 1, 2, true>(a, stream);
    if (ns == 2 && nd == 1) return launch_attn_inst<2, 1, 1, true>(a, stream);
    if (ns == 1 && nd == 2) return launch_attn_inst<1, 2, 1, true>(a, stream);
    return launch_attn_inst<2, 2, 1, true>(a, stream);
}

}  // namespace nrms
