// Split-bf16 ("bf16x3") and plain bf16 GEMMs on v_mfma_f32_16x16x32_bf16.
//
// fp32 operands are written x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significant bits
// together).  A product is accumulated in fp32 as  hi_a*hi_b + hi_a*lo_b + lo_a*hi_b  (the lo*lo term,
// 2^-16 relative, is dropped): three bf16 MFMAs per fp32-equivalent MFMA step, i.e. 3/16 of the
// f32-MFMA time at ~2^-16 relative error per product -- two orders of magnitude inside the 1e-4 score
// bar.  NPASS = 1 keeps only hi*hi (plain bf16 inputs, fp32 accumulate).
//
// HBM tensors stay fp32: weights are split ONCE per call into bf16 hi/lo planes by split_planes_kernel
// (they are < 1.1 MB and re-read by every row tile), activations are split while being staged into
// LDS (16 VALU ops per thread and stage, negligible next to the gather/RNG work that was moved out of
// the GEMM waves -- see gemm.h).
#include <stdlib.h>

#include "gemm.h"

namespace nrms {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BF_BM = 256;            // rows per workgroup tile (8 waves x 32)
constexpr int BF_BK = 32;             // bf16 elements of K per stage = one 64-byte row segment
constexpr int BF_THREADS = 512;
constexpr int BF_ROWB = 64;           // bytes per staged row (32 bf16)

struct Split8 { bf16x8 hi, lo; };

__device__ __forceinline__ Split8 split8(const f32x4& u, const f32x4& v) {
    Split8 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h0 = (__bf16)u[i], h1 = (__bf16)v[i];
        s.hi[i] = h0; s.hi[4 + i] = h1;
        s.lo[i] = (__bf16)(u[i] - (float)h0);
        s.lo[4 + i] = (__bf16)(v[i] - (float)h1);
    }
    return s;
}

// w [rows, cols] fp32  ->  hi, lo [rows, cols_p] bf16, zero padded to cols_p (a multiple of 32) and
// K-PERMUTED inside every 32-column block: source column t = 16 u + 4 kq + e (u < 2, kq < 4, e < 4) is
// stored at 8 kq + 4 u + e.  The 16-byte chunk kq of a block then holds exactly the 8 k-values
// {4 kq + e, 16 + 4 kq + e} that lane group kq of the MFMA consumes, and the matching A fragment is
// two float4 global loads that are each 64 contiguous bytes per row across the four lane groups.
__global__ void split_planes_kernel(const float* w, int rows, int rows_p, int cols, int cols_p, __bf16* hi, __bf16* lo) {
    // K-BLOCK-MAJOR planes [cols_p/32][rows_p][32]: the weight tile of one K stage (all rows, 32 k) is a
    // single contiguous region, so staging it is 1-KiB-contiguous wave loads (8 full 128-byte lines per
    // instruction) instead of sixteen 64-byte row fragments -- the GEMMs are bound by the vector-memory
    // request rate, not by bytes.  Rows >= `rows` and columns >= `cols` are zero.
    const long total = (long)rows_p * cols_p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols_p;
        const int c = (int)(i - r * cols_p);
        const float x = (r < rows && c < cols) ? w[r * cols + c] : 0.f;
        const __bf16 h = (__bf16)x;
        const int t = c & 31, u = t >> 4, kq = (t >> 2) & 3, e = t & 3;
        const long dst = ((long)(c >> 5) * rows_p + r) * 32 + 8 * kq + 4 * u + e;
        hi[dst] = h;
        lo[dst] = (__bf16)(x - (float)h);
    }
}

int launch_split_planes(const float* w, int rows, int rows_p, int cols, int cols_p, void* hi, void* lo,
                        hipStream_t stream) {
    const long total = (long)rows_p * cols_p;
    TimingScope ts("split_planes", stream);
    hipLaunchKernelGGL(split_planes_kernel, dim3(cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256)), dim3(256), 0, stream,
                       w, rows, rows_p, cols, cols_p, (__bf16*)hi, (__bf16*)lo);
    return check_launch("split_planes");
}

struct BFArgs {
    NTArgs g;                 // same meaning as the fp32 kernel; g.W is unused
    const __bf16* whi;        // [Kp/32][Np][32] bf16, zero padded, K-permuted inside each block
    const __bf16* wlo;        // same (NPASS == 3)
    int Kp, Np;
};

// Main loop.  Only the weight tile is shared between waves, so only it goes through LDS
// ([16 NT rows][64 B] per plane, 16-byte chunks XOR-swizzled as in the fp32 loop, 2 stages).  The A
// rows of a wave are private to it: each lane loads its own MFMA fragment straight from HBM (two
// float4 = k {4kq..4kq+3} and {16+4kq..}, see split_planes_kernel), two stages ahead, and splits it
// into bf16 hi/lo in registers.  One barrier per stage (for the weight buffers).
template <int NT, int AMODE, int NPASS>
__device__ __forceinline__ void bf16_nt_mainloop(const BFArgs& a, int row0, int rows_valid, int col0,
                                                 f32x4 (&acc)[2][NT], char* smem) {
    constexpr int P = NPASS == 3 ? 2 : 1;
    constexpr int B_PLANE = NT * 16 * BF_ROWB;
    constexpr int STAGE = P * B_PLANE;
    constexpr int B_CH = NT * 16 * 4;                      // 16-byte chunks per B plane
    constexpr int B_IT = (B_CH + BF_THREADS - 1) / BF_THREADS;
    const NTArgs& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int chunk = tid & 3;

    // ---- this lane's two A rows (one per 16-row MFMA tile).  Rows past the tile's end alias its last
    // valid row: their accumulators are computed and never stored, and every load stays unconditional.
    // hipcc turns a conditional load into an exec-mask branch per load and, worse, waits (vmcnt 0) where
    // the loaded value is first TOUCHED -- a zero-select or the tanh' factor next to the load therefore
    // serialises the whole "prefetch" behind the memory latency.  Raw values only, selects at the split.
    const float* arow[2];
    float ascale[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long gr = (long)row0 + min(32 * wave + 16 * mt + r16, rows_valid - 1);
        if (AMODE == A_PLAIN) { arow[mt] = g.A + (g.a_rows != nullptr ? (long)g.a_rows[gr] : gr) * g.lda; ascale[mt] = 0.f; }
        else { arow[mt] = g.T + gr * (long)g.K; ascale[mt] = g.ds[gr]; }
    }
    long b_src[B_IT];
    int b_off[B_IT];
    bool b_zero[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + BF_THREADS * i;
        const int r = idx >> 2, n = col0 + r;
        b_src[i] = (long)min(n, a.Np - 1) * 32 + 8 * chunk;     // clamped: always a valid plane address
        b_zero[i] = n >= a.Np;                                  // (rows N..Np-1 are zero in the planes)
        b_off[i] = idx < B_CH ? (r * 4 + (chunk ^ nt_swz(r))) * 16 : -1;
    }
    const int b_frag = r16 * BF_ROWB + (kq ^ nt_swz(r16)) * 16;

    f32x4 ar[2][2], qr[2];             // raw A fragment of the NEXT stage [m tile][k half] (+ q slice, A_DZ)
    bf16x8 bst[B_IT];                  // weight staging registers: ONE plane at a time (see the loop)
    const int k_last = g.K - 4;        // K % 4 == 0 (checked by the launcher)
    auto load_a = [&](int k0) {
        const int k = k0 + 4 * kq;
        const int ka = min(k, k_last), kb = min(k + 16, k_last);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            ar[mt][0] = *reinterpret_cast<const f32x4*>(arow[mt] + ka);
            ar[mt][1] = *reinterpret_cast<const f32x4*>(arow[mt] + kb);
        }
        if (AMODE == A_DZ) {
            qr[0] = *reinterpret_cast<const f32x4*>(g.qv + ka);
            qr[1] = *reinterpret_cast<const f32x4*>(g.qv + kb);
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto split_a = [&](int k0, Split8& a0, Split8& a1) {
        const int k = k0 + 4 * kq;
        const bool ok0 = k < g.K, ok1 = k + 16 < g.K;
        Split8 out[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 u = ar[mt][0], v = ar[mt][1];
            if (AMODE == A_DZ) {
                u = ascale[mt] * qr[0] * (1.0f - u * u);
                v = ascale[mt] * qr[1] * (1.0f - v * v);
            }
            out[mt] = split8(ok0 ? u : zero4, ok1 ? v : zero4);
        }
        a0 = out[0]; a1 = out[1];
    };
    auto load_b = [&](const __bf16* plane, int k0) {
        const __bf16* blk = plane + (long)(k0 >> 5) * a.Np * 32;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) bst[i] = *reinterpret_cast<const bf16x8*>(blk + b_src[i]);
    };
    auto store_b = [&](char* st) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if ((i + 1) * BF_THREADS <= B_CH || b_off[i] >= 0)
                *reinterpret_cast<bf16x8*>(st + b_off[i]) = b_zero[i] ? z : bst[i];
    };
    auto read_b = [&](const char* cur, int nt, bf16x8& hi, bf16x8& lo) {
        hi = *reinterpret_cast<const bf16x8*>(cur + b_frag + nt * 16 * BF_ROWB);
        if (NPASS == 3) lo = *reinterpret_cast<const bf16x8*>(cur + b_frag + B_PLANE + nt * 16 * BF_ROWB);
    };
    auto mma = [&](int nt, const Split8& a0, const Split8& a1, const bf16x8& bhi, const bf16x8& blo) {
        if (NPASS == 3) {
            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.hi, blo, acc[0][nt], 0, 0, 0);
            acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.hi, blo, acc[1][nt], 0, 0, 0);
            acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.lo, bhi, acc[0][nt], 0, 0, 0);
            acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.lo, bhi, acc[1][nt], 0, 0, 0);
        }
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.hi, bhi, acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.hi, bhi, acc[1][nt], 0, 0, 0);
    };

    const int n_stage = (g.K + BF_BK - 1) / BF_BK;
    load_b(a.whi, 0);
    store_b(smem);
    if (NPASS == 3) { load_b(a.wlo, 0); store_b(smem + B_PLANE); }
    load_a(0);
    __syncthreads();
    constexpr int HALF = (NT + 1) / 2;
#pragma unroll 2
    for (int s = 0; s < n_stage; ++s) {
        const char* cur = smem + (s & 1) * STAGE;
        char* nxt = smem + ((s + 1) & 1) * STAGE;
        const bool more = s + 1 < n_stage;
        const int kn = (s + 1) * BF_BK;
        // the raw registers die at the split, so the next stage's fragment is loaded straight back into
        // them and stays in flight for the whole stage (no slot array: a run-time slot index makes hipcc
        // address the registers through movrel, behind a full vmcnt(0))
        Split8 a0, a1;
        split_a(s * BF_BK, a0, a1);
        if (more) {
            load_b(a.whi, kn);                                 // hi plane of the next stage flies during half 1
            load_a(kn);
        }
        // fragment reads run one tile ahead of the MFMAs that consume them (6 MFMAs = 96 cycles per
        // tile would otherwise expose the full LDS latency every iteration)
        bf16x8 bh0, bl0, bh1, bl1;
        read_b(cur, 0, bh0, bl0);
#pragma unroll
        for (int nt = 0; nt < HALF; nt += 2) {
            if (nt + 1 < NT) read_b(cur, nt + 1, bh1, bl1);
            mma(nt, a0, a1, bh0, bl0);
            if (nt + 1 < HALF) {
                if (nt + 2 < NT) read_b(cur, nt + 2, bh0, bl0);
                mma(nt + 1, a0, a1, bh1, bl1);
            }
        }
        if (more) {
            store_b(nxt);
            if (NPASS == 3) load_b(a.wlo, kn);                 // lo plane flies during half 2
        }
        // second half continues the same ping-pong (HALF even -> next tile sits in bh0, odd -> bh1)
        if (HALF % 2 == 0) {
#pragma unroll
            for (int nt = HALF; nt < NT; nt += 2) {
                if (nt + 1 < NT) read_b(cur, nt + 1, bh1, bl1);
                mma(nt, a0, a1, bh0, bl0);
                if (nt + 1 < NT) {
                    if (nt + 2 < NT) read_b(cur, nt + 2, bh0, bl0);
                    mma(nt + 1, a0, a1, bh1, bl1);
                }
            }
        } else {
#pragma unroll
            for (int nt = HALF; nt < NT; nt += 2) {
                if (nt + 1 < NT) read_b(cur, nt + 1, bh0, bl0);
                mma(nt, a0, a1, bh1, bl1);
                if (nt + 1 < NT) {
                    if (nt + 2 < NT) read_b(cur, nt + 2, bh1, bl1);
                    mma(nt + 1, a0, a1, bh0, bl0);
                }
            }
        }
        if (more && NPASS == 3) store_b(nxt + B_PLANE);
        __syncthreads();
    }
}

template <int NT, int AMODE, int EMODE, int NPASS>
__global__ __launch_bounds__(BF_THREADS, 2) void gemm_nt_bf16_kernel(BFArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NTArgs& g = a.g;
    const int n_ct = (g.N + NT * 16 - 1) / (NT * 16);
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int rt = (jb / n_ct) * 8 + xcd;
    const int row0 = rt * g.rows_per_tile;
    const int M = g.m_dev != nullptr ? *g.m_dev : g.M;
    if (row0 >= M) return;
    const int rows_valid = min(g.rows_per_tile, M - row0);
    const int col0 = (jb % n_ct) * (NT * 16);

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16_nt_mainloop<NT, AMODE, NPASS>(a, row0, rows_valid, col0, acc, smem);

    // ---- epilogue (same C layout as the fp32 16x16x4 MFMA); the weight buffers are dead: reuse them
    // (launch_bf_inst sizes the dynamic LDS as max(2 stages, 8 waves x strip))
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nt_epilogue<NT, EMODE>(g, acc, row0, rows_valid, col0, wave, lane,
                           reinterpret_cast<float*>(smem) + wave * 8 * (16 * NT + 8));
}

template <int NT, int AMODE, int EMODE, int NPASS>
static int launch_bf_inst(const BFArgs& a, hipStream_t stream, const char* name) {
    constexpr int P = NPASS == 3 ? 2 : 1;
    constexpr size_t stage_bytes = 2 * (size_t)P * (NT * 16) * BF_ROWB;
    constexpr size_t strip_bytes = (size_t)8 * 8 * (16 * NT + 8) * sizeof(float);
    constexpr size_t lds_bytes = stage_bytes > strip_bytes ? stage_bytes : strip_bytes;
    const void* fn = (const void*)gemm_nt_bf16_kernel<NT, AMODE, EMODE, NPASS>;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return NRMS_ELAUNCH; }
    dim3 grid(cdiv(cdiv(a.g.M, a.g.rows_per_tile), 8) * 8 * cdiv(a.g.N, NT * 16));
    TimingScope ts(name, stream);
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<NT, AMODE, EMODE, NPASS>), grid, dim3(BF_THREADS), lds_bytes, stream, a);
    return check_launch(name);
}

template <int AMODE, int EMODE, int NPASS>
static int launch_bf_mode(const BFArgs& a, hipStream_t stream, const char* name) {
    const int N = a.g.N;
    // column-tile width: least padding among wide tiles (narrow tiles re-read A once per tile)
    const int cand[6] = {19, 15, 13, 10, 8, 4};
    int nt = 19;
    long best = -1;
    for (int i = 0; i < 6; ++i) {
        if (cand[i] < 8 && N > 128) continue;
        const long pad = (long)cdiv(N, cand[i] * 16) * cand[i] * 16;
        if (best < 0 || pad < best) { best = pad; nt = cand[i]; }
    }
    switch (nt) {
        case 19: return launch_bf_inst<19, AMODE, EMODE, NPASS>(a, stream, name);
        case 15: return launch_bf_inst<15, AMODE, EMODE, NPASS>(a, stream, name);
        case 13: return launch_bf_inst<13, AMODE, EMODE, NPASS>(a, stream, name);
        case 10: return launch_bf_inst<10, AMODE, EMODE, NPASS>(a, stream, name);
        case 8: return launch_bf_inst<8, AMODE, EMODE, NPASS>(a, stream, name);
        default: return launch_bf_inst<4, AMODE, EMODE, NPASS>(a, stream, name);
    }
}

// W planes: hi at wplanes, lo at wplanes + N*Kp (bf16 elements); Kp = K rounded up to 32.
int launch_gemm_nt_bf16(int amode, int emode, int npass, const NTArgs& g_in, void* wplanes, hipStream_t stream,
                        const char* name) {
    if (g_in.M <= 0) return NRMS_OK;
    if ((g_in.K & 3) != 0) { set_error("%s: K=%d must be a multiple of 4", name, g_in.K); return NRMS_EINVAL; }
    BFArgs a;
    a.g = g_in;
    a.g.rows_per_tile = BF_BM;
    a.Kp = cdiv(g_in.K, BF_BK) * BF_BK;
    a.Np = cdiv(g_in.N, 16) * 16;
    a.whi = (const __bf16*)wplanes;
    a.wlo = a.whi + (long)a.Np * a.Kp;
    int rc = launch_split_planes(g_in.W, g_in.N, a.Np, g_in.K, a.Kp, wplanes, (void*)a.wlo, stream);
    if (rc) return rc;
    if (npass == 3) {
        if (amode == A_PLAIN && emode == E_STORE) return launch_bf_mode<A_PLAIN, E_STORE, 3>(a, stream, name);
        if (amode == A_DZ && emode == E_DCTX) return launch_bf_mode<A_DZ, E_DCTX, 3>(a, stream, name);
        if (amode == A_DZ && emode == E_STORE) return launch_bf_mode<A_DZ, E_STORE, 3>(a, stream, name);
    } else {
        if (amode == A_PLAIN && emode == E_STORE) return launch_bf_mode<A_PLAIN, E_STORE, 1>(a, stream, name);
        if (amode == A_DZ && emode == E_DCTX) return launch_bf_mode<A_DZ, E_DCTX, 1>(a, stream, name);
        if (amode == A_DZ && emode == E_STORE) return launch_bf_mode<A_DZ, E_STORE, 1>(a, stream, name);
    }
    set_error("%s: unsupported bf16 gemm mode %d/%d", name, amode, emode);
    return NRMS_EINVAL;
}

// =======================================================================================
// Additive-attention forward on the split-bf16 main loop (AdditiveAttention, model/nrms_v0.py:100-126):
// one workgroup owns floor(256/S) whole sequences; T = tanh(C Wa^T + ba) from the MFMA accumulators,
// s = T.q reduced across the tile's columns in registers, T written in whole rows through the LDS
// strips, then softmax over each sequence and the weighted pooling of the (L2-resident) C rows.
// =======================================================================================
template <int NT, int NPASS>
__global__ __launch_bounds__(BF_THREADS, 2) void addattn_fwd_bf16_kernel(AddFwdArgs a, BFArgs b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float sc[BF_BM];
    const NTArgs& g = b.g;
    const int row0 = blockIdx.x * g.rows_per_tile;
    const int rows_valid = min(g.rows_per_tile, g.M - row0);
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16_nt_mainloop<NT, A_PLAIN, NPASS>(b, row0, rows_valid, 0, acc, smem);

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
            float part = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float t = fast_tanh(acc[mt][nt][reg] + bcol[nt]);
                part += t * qcol[nt];
                acc[mt][nt][reg] = t;               // (columns >= N are never stored)
            }
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            part += __shfl_xor(part, 8, 64);
            if (r16 == 0) sc[32 * wave + 16 * mt + 4 * kq + reg] = part;
        }
    if (a.T != nullptr) {
        NTArgs tg = g;                      // T [M, q]: plain coalesced store of the tanh tile
        tg.bias = nullptr; tg.C = a.T; tg.ldc = g.N;
        nt_epilogue<NT, E_STORE>(tg, acc, row0, rows_valid, 0, wave, lane,
                                 reinterpret_cast<float*>(smem) + wave * 8 * (16 * NT + 8));
    }
    __syncthreads();
    // softmax over each sequence: one 64-lane wave per sequence, a lane per position (S <= 64), reductions by
    // shuffles (the first version let `spb` single threads walk their sequences serially while 500 others idled)
    const int spb = rows_valid / a.S;
    for (int sq = wave; sq < spb; sq += BF_THREADS / 64) {
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
    for (int idx = tid; idx < spb * (a.d / 4); idx += BF_THREADS) {
        const int sq = idx / (a.d / 4), c4 = idx - sq * (a.d / 4);
        const float* crow = g.A + ((long)row0 + sq * a.S) * g.lda + 4 * c4;
        const float* w = sc + sq * a.S;
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
        int i = 0;
        for (; i + 1 < a.S; i += 2) {               // two independent row loads per trip (L2-resident tile)
            o0 += w[i] * *reinterpret_cast<const f32x4*>(crow + (long)i * g.lda);
            o1 += w[i + 1] * *reinterpret_cast<const f32x4*>(crow + (long)(i + 1) * g.lda);
        }
        if (i < a.S) o0 += w[i] * *reinterpret_cast<const f32x4*>(crow + (long)i * g.lda);
        *reinterpret_cast<f32x4*>(a.out + (seq0 + sq) * a.d + 4 * c4) = o0 + o1;
    }
}

template <int NT, int NPASS>
static int launch_addfwd_bf_inst(const AddFwdArgs& a, const BFArgs& b, hipStream_t stream) {
    constexpr int P = NPASS == 3 ? 2 : 1;
    constexpr size_t stage_bytes = 2 * (size_t)P * (NT * 16) * BF_ROWB;
    constexpr size_t strip_bytes = (size_t)8 * 8 * (16 * NT + 8) * sizeof(float);
    constexpr size_t lds_bytes = stage_bytes > strip_bytes ? stage_bytes : strip_bytes;
    const void* fn = (const void*)addattn_fwd_bf16_kernel<NT, NPASS>;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("addattn_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return NRMS_ELAUNCH; }
    TimingScope ts("addattn_fwd", stream);
    hipLaunchKernelGGL((addattn_fwd_bf16_kernel<NT, NPASS>), dim3(cdiv(b.g.M, b.g.rows_per_tile)), dim3(BF_THREADS),
                       lds_bytes, stream, a, b);
    return check_launch("addattn_fwd");
}

int launch_addattn_fwd_bf16(int npass, const AddFwdArgs& a, void* wplanes, hipStream_t stream) {
    BFArgs b;
    b.g = a.g;
    b.g.rows_per_tile = (BF_BM / a.S) * a.S;
    b.Kp = cdiv(a.g.K, BF_BK) * BF_BK;
    b.Np = cdiv(a.g.N, 16) * 16;
    b.whi = (const __bf16*)wplanes;
    b.wlo = b.whi + (long)b.Np * b.Kp;
    int rc = launch_split_planes(a.g.W, a.g.N, b.Np, a.g.K, b.Kp, wplanes, (void*)b.wlo, stream);
    if (rc) return rc;
    const int q = a.g.N;
    if (npass == 3) {
        if (q <= 64) return launch_addfwd_bf_inst<4, 3>(a, b, stream);
        if (q <= 128) return launch_addfwd_bf_inst<8, 3>(a, b, stream);
        return launch_addfwd_bf_inst<13, 3>(a, b, stream);
    }
    if (q <= 64) return launch_addfwd_bf_inst<4, 1>(a, b, stream);
    if (q <= 128) return launch_addfwd_bf_inst<8, 1>(a, b, stream);
    return launch_addfwd_bf_inst<13, 1>(a, b, stream);
}

// =======================================================================================
// TN in split-bf16: dW[N,K(+1)] = sum_m A'[m,N]^T B'[m,K(+ones)]
// =======================================================================================
// Same decomposition as the fp32 kernel (gemm.hip): 8 waves = 4 (N) x 2 (K), NTN x NTK tiles of
// 16x16 per wave, 32 rows of M per stage, M split over workgroups into partial slabs.  Here a stage
// is exactly ONE v_mfma_f32_16x16x32_bf16 k-step: the staged fp32 rows are split into bf16 hi/lo
// planes [32 m][cols] (row-major, as they arrive), and both MFMA operands -- which need 8
// consecutive m for one column -- come from ds_read_b64_tr_b16, the hardware transposed read (a
// 16-lane group reads 4 rows x 16 columns and returns them column-major).
//   k <-> m assignment inside a stage: lane group kq (= lane>>4) takes m = 4 kq + j (j < 4) and
//   16 + 4 kq + (j-4): then one 32-lane half reads rows {0..7} (or {16..23}) and, with a row pitch
//   of 8*odd dwords, its 32 eight-byte accesses tile the 64 banks exactly (conflict-free).
constexpr int TB_WN = 4, TB_WK = 2, TB_MC = 32, TB_THREADS = 512;
constexpr int TB_AW = 320, TB_BW = 160;                 // staged columns (max)
constexpr int TB_PA = 336, TB_PB = 176;                 // row pitch in bf16 elements: 168 / 88 dwords = 8*odd
constexpr int TB_A_PLANE = TB_MC * TB_PA * 2, TB_B_PLANE = TB_MC * TB_PB * 2;     // bytes
constexpr int TB_A_IT = (TB_MC * TB_AW / 4) / TB_THREADS;                           // 5
constexpr int TB_B_IT = (TB_MC * TB_BW / 4 + TB_THREADS - 1) / TB_THREADS;          // 3

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct TNGeomB { int n_tiles, k_tiles, n_wg, k_wg, n_tpw, k_tpw; };

static TNGeomB tnb_geom(int N, int K) {
    TNGeomB g;
    g.n_tiles = cdiv(N, 16);
    g.k_tiles = cdiv(K + 1, 16);
    g.n_wg = cdiv(g.n_tiles, TB_WN * 5);
    g.k_wg = cdiv(g.k_tiles, TB_WK * 5);
    // prefer a tile count that divides an XCD's 32 CUs (one full round); give up if none is near
    const int n0 = g.n_wg;
    for (int n = n0; n <= 2 * n0 + 1 && n <= g.n_tiles; ++n)
        if (n * g.k_wg <= 32 && 32 % (n * g.k_wg) == 0) { g.n_wg = n; break; }
    g.n_tpw = cdiv(g.n_tiles, g.n_wg);
    g.k_tpw = cdiv(g.k_tiles, g.k_wg);
    return g;
}

__device__ __forceinline__ bf16x8 tr_frag(const char* plane, int pitch_b, int col0, int lane) {
    // operand fragment for columns col0..col0+15: element j of lane (c = lane&15, kq = lane>>4)
    // = image[m(kq, j)][col0 + c]
    const int l16 = lane & 15, kq = lane >> 4;
    const int q = l16 >> 2, p = l16 & 3;
    const char* a0 = plane + (4 * kq + q) * pitch_b + (col0 + 4 * p) * 2;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 16 * pitch_b));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = lo4[i]; r[4 + i] = hi4[i]; }
    return __builtin_bit_cast(bf16x8, r);
}

template <int AMODE, int NTN, int NTK, int NPASS>
__global__ __launch_bounds__(TB_THREADS, 2) void gemm_tn_bf16_kernel(TNArgs a, TNGeomB g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int P = NPASS == 3 ? 2 : 1;
    constexpr int STAGE = P * (TB_A_PLANE + TB_B_PLANE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;

    const int out_wgs = g.n_wg * g.k_wg;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile = jb % out_wgs;
    const int split = (jb / out_wgs) * 8 + xcd;
    if (split >= a.splits) return;
    const int bx = tile % g.n_wg, by = tile / g.n_wg;
    const int nt0 = bx * g.n_tpw, nt_cnt = max(0, min(g.n_tpw, g.n_tiles - nt0));
    const int kt0 = by * g.k_tpw, kt_cnt = max(0, min(g.k_tpw, g.k_tiles - kt0));
    const int n_q = (nt_cnt + TB_WN - 1) / TB_WN, k_h = (kt_cnt + TB_WK - 1) / TB_WK;
    const int my_nt0 = wn * n_q, my_ntn = max(0, min(n_q, nt_cnt - my_nt0));
    const int my_kt0 = wk * k_h, my_ktn = max(0, min(k_h, kt_cnt - my_kt0));
    const int n_cols = nt_cnt * 16, k_cols = kt_cnt * 16;
    const int ncol0 = nt0 * 16, kcol0 = kt0 * 16;
    const int M = a.m_dev != nullptr ? *a.m_dev : a.M;
    const int rps = a.m_dev != nullptr ? cdiv(cdiv(M, a.splits), TB_MC) * TB_MC : a.rows_per_split;
    const int m_begin = split * rps;
    const int m_end = min(M, m_begin + rps);

    // Staging slots of this thread.  Every global load is unconditional from a clamped (valid) address;
    // validity is applied when the registers are split into the LDS planes, a whole stage later (a
    // predicate next to the load costs an exec-mask branch per load and drags the wait up to it).
    int a_r[TB_A_IT], a_c[TB_A_IT], a_n[TB_A_IT], b_r[TB_B_IT], b_c[TB_B_IT], b_k[TB_B_IT];
    bool a_ok[TB_A_IT], b_ok[TB_B_IT], b_one[TB_B_IT];
    f32x4 qv_r[TB_A_IT];
#pragma unroll
    for (int i = 0; i < TB_A_IT; ++i) {
        const int sl = tid + TB_THREADS * i;
        a_r[i] = sl / (TB_AW / 4);
        a_c[i] = (sl - a_r[i] * (TB_AW / 4)) * 4;
        a_ok[i] = a_c[i] < n_cols && ncol0 + a_c[i] < a.N;
        a_n[i] = min(ncol0 + a_c[i], a.N - 4);
        if (AMODE == A_DZ) qv_r[i] = *reinterpret_cast<const f32x4*>(a.qv + a_n[i]);
    }
#pragma unroll
    for (int i = 0; i < TB_B_IT; ++i) {
        const int sl = tid + TB_THREADS * i;
        b_r[i] = sl / (TB_BW / 4);
        b_c[i] = (sl - b_r[i] * (TB_BW / 4)) * 4;
        if (b_r[i] >= TB_MC) { b_r[i] = 0; b_c[i] = TB_BW; }
        const int k = kcol0 + b_c[i];
        b_ok[i] = b_c[i] < k_cols && k < a.K;
        b_one[i] = b_c[i] < k_cols && k == a.K;            // the ones column that yields the bias gradient
        b_k[i] = min(k, a.K - 4);
    }

    struct Regs { f32x4 a[TB_A_IT], b[TB_B_IT]; float ds[TB_A_IT]; };
    auto load_stage = [&](Regs& R, int m0) {
#pragma unroll
        for (int i = 0; i < TB_A_IT; ++i) {
            const long m = min(m0 + a_r[i], m_end - 1);
            if (AMODE == A_PLAIN) {
                R.a[i] = *reinterpret_cast<const f32x4*>(a.A + m * a.lda + a_n[i]);
            } else {
                R.a[i] = *reinterpret_cast<const f32x4*>(a.T + m * (long)a.N + a_n[i]);
                R.ds[i] = a.ds[m];
            }
        }
#pragma unroll
        for (int i = 0; i < TB_B_IT; ++i) {
            const long m = min(m0 + b_r[i], m_end - 1);
            R.b[i] = *reinterpret_cast<const f32x4*>(a.B + m * a.ldb + b_k[i]);
        }
    };
    auto split4 = [](const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 h = (__bf16)v[e];
            hi[e] = h;
            lo[e] = (__bf16)(v[e] - (float)h);
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto store_stage = [&](const Regs& R, int m0, char* st) {
#pragma unroll
        for (int i = 0; i < TB_A_IT; ++i) {
            f32x4 v = R.a[i];
            if (AMODE == A_DZ) v = R.ds[i] * qv_r[i] * (1.0f - v * v);
            v = (a_ok[i] && m0 + a_r[i] < m_end) ? v : zero4;
            bf16x4 hi, lo;
            split4(v, hi, lo);
            const int off = (a_r[i] * TB_PA + a_c[i]) * 2;
            *reinterpret_cast<bf16x4*>(st + off) = hi;
            if (NPASS == 3) *reinterpret_cast<bf16x4*>(st + TB_A_PLANE + off) = lo;
        }
#pragma unroll
        for (int i = 0; i < TB_B_IT; ++i) {
            if (b_c[i] >= TB_BW) continue;
            const bool mok = m0 + b_r[i] < m_end;
            f32x4 v = (b_ok[i] && mok) ? R.b[i] : zero4;
            if (b_one[i] && mok) v[0] = 1.0f;
            bf16x4 hi, lo;
            split4(v, hi, lo);
            const int off = P * TB_A_PLANE + (b_r[i] * TB_PB + b_c[i]) * 2;
            *reinterpret_cast<bf16x4*>(st + off) = hi;
            if (NPASS == 3) *reinterpret_cast<bf16x4*>(st + TB_B_PLANE + off) = lo;
        }
    };

    f32x4 acc[NTN][NTK];
#pragma unroll
    for (int i = 0; i < NTN; ++i)
#pragma unroll
        for (int jj = 0; jj < NTK; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int s) {
        const char* As = smem + (s & 1) * STAGE;
        const char* Bs = As + P * TB_A_PLANE;
        bf16x8 bh[NTK], bl[NTK];
#pragma unroll
        for (int jj = 0; jj < NTK; ++jj) {
            bh[jj] = tr_frag(Bs, TB_PB * 2, (my_kt0 + jj) * 16, lane);
            if (NPASS == 3) bl[jj] = tr_frag(Bs + TB_B_PLANE, TB_PB * 2, (my_kt0 + jj) * 16, lane);
        }
#pragma unroll
        for (int i = 0; i < NTN; ++i) {
            const bf16x8 ah = tr_frag(As, TB_PA * 2, (my_nt0 + i) * 16, lane);
            bf16x8 al;
            if (NPASS == 3) al = tr_frag(As + TB_A_PLANE, TB_PA * 2, (my_nt0 + i) * 16, lane);
#pragma unroll
            for (int jj = 0; jj < NTK; ++jj) {
                if (NPASS == 3) {
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[jj], acc[i][jj], 0, 0, 0);
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[jj], acc[i][jj], 0, 0, 0);
                }
                acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[jj], acc[i][jj], 0, 0, 0);
            }
        }
    };

    // A stage is only ~1k cycles of MFMA per wave, less than one HBM round trip, and all 8 waves move
    // in lockstep: where the accumulators leave room the raw rows are prefetched TWO stages ahead in
    // two register sets (named statically: the loop is unrolled by two, a run-time set index would
    // go through movrel).  At the top of stage s the set holding stage s+1 (issued a whole stage
    // ago) is split into the free LDS buffer while the loads of stage s+2 are already in flight.
    // The stage body is branch-free on purpose: loads past the split's end read its last row (clamped)
    // and are zeroed at the split, the odd stage count is rounded up to even (one all-zero stage).
    // With an `if (more)` around the loads, hipcc's waitcnt pass merges the two paths and then waits
    // for the loads it has JUST issued (vmcnt(7) after 8 loads) -- again no prefetch.
    constexpr bool DEEP = NTN * NTK <= 20;
    int n_stage = (m_end - m_begin + TB_MC - 1) / TB_MC;
    Regs R0, R1;
    if (n_stage > 0) {                 // an empty split (compacted M) still writes its all-zero slab below
    load_stage(R1, m_begin);
    store_stage(R1, m_begin, smem);
    if (DEEP) {
        n_stage = (n_stage + 1) & ~1;
        load_stage(R0, m_begin + TB_MC);
        __syncthreads();
        auto body = [&](const Regs& Rs, Regs& Rl, int s) {
            load_stage(Rl, m_begin + (s + 2) * TB_MC);
            store_stage(Rs, m_begin + (s + 1) * TB_MC, smem + ((s + 1) & 1) * STAGE);
            compute(s);
            __syncthreads();
        };
        for (int s = 0; s < n_stage; s += 2) {
            body(R0, R1, s);
            body(R1, R0, s + 1);
        }
    } else {
        __syncthreads();
        for (int s = 0; s < n_stage; ++s) {
            load_stage(R0, m_begin + (s + 1) * TB_MC);
            compute(s);
            store_stage(R0, m_begin + (s + 1) * TB_MC, smem + ((s + 1) & 1) * STAGE);
            __syncthreads();
        }
    }
    }
    const long n_pad = (long)g.n_tiles * 16, k_pad = (long)g.k_tiles * 16;
    float* slab = a.partial + (long)split * n_pad * k_pad;
#pragma unroll
    for (int i = 0; i < NTN; ++i) {
        if (i >= my_ntn) continue;
#pragma unroll
        for (int jj = 0; jj < NTK; ++jj) {
            if (jj >= my_ktn) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long n = ncol0 + (my_nt0 + i) * 16 + 4 * kq + reg;
                const long k = kcol0 + (my_kt0 + jj) * 16 + r16;
                slab[n * k_pad + k] = acc[i][jj][reg];
            }
        }
    }
}

__global__ void tnb_reduce_kernel(const float* partial, int splits, int N, int K, long n_pad, long k_pad, float* dW,
                                  float* dbias, HeadPerm perm) {
    const long total = (long)N * (K + 1);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx / (K + 1), k = idx - n * (K + 1);
        // fixed summation order, four independent partial sums so that four slab reads are in flight
        const float* pp = partial + n * k_pad + k;
        const long slab = n_pad * k_pad;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sp = 0;
        for (; sp + 3 < splits; sp += 4) {
            s0 += pp[(long)sp * slab];
            s1 += pp[(long)(sp + 1) * slab];
            s2 += pp[(long)(sp + 2) * slab];
            s3 += pp[(long)(sp + 3) * slab];
        }
        for (; sp < splits; ++sp) s0 += pp[(long)sp * slab];
        const float s = (s0 + s1) + (s2 + s3);
        const long nd = perm.src((int)n);
        if (k < K) dW[nd * K + k] += s;
        else if (dbias != nullptr) dbias[nd] += s;
    }
}

static int tnb_splits(int M, const TNGeomB& g) {
    const int out_wgs = g.n_wg * g.k_wg;
    int splits = 8 * (out_wgs <= 32 ? 32 / out_wgs : 1);
    const int max_splits = cdiv(M, 8 * TB_MC);
    if (splits > max_splits) splits = max_splits;
    return splits < 1 ? 1 : splits;
}

// same partial-slab workspace as the fp32 kernel (gemm_tn_workspace_floats covers it)
int launch_gemm_tn_bf16(int npass, const TNArgs& a_in, hipStream_t stream, const char* name) {
    if (a_in.M <= 0) return NRMS_OK;
    TNArgs a = a_in;
    const TNGeomB g = tnb_geom(a.N, a.K);
    if ((a.N & 3) != 0 || (a.K & 3) != 0) { set_error("%s: N,K must be multiples of 4", name); return NRMS_EINVAL; }
    a.splits = tnb_splits(a.M, g);
    a.rows_per_split = cdiv(cdiv(a.M, a.splits), TB_MC) * TB_MC;
    a.splits = cdiv(a.M, a.rows_per_split);
    dim3 grid(cdiv(a.splits, 8) * 8 * g.n_wg * g.k_wg);
    const int n_q = cdiv(g.n_tpw, TB_WN), k_h = cdiv(g.k_tpw, TB_WK);
    int rc = NRMS_OK;
#define TNB_LAUNCH(MODE, NN, KK, NP)                                                                                \
    do {                                                                                                            \
        constexpr size_t lds_bytes = 2 * (size_t)(NP == 3 ? 2 : 1) * (TB_A_PLANE + TB_B_PLANE);                     \
        const void* fn = (const void*)gemm_tn_bf16_kernel<MODE, NN, KK, NP>;                                        \
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   \
        if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return NRMS_ELAUNCH; } \
        TimingScope ts(name, stream);                                                                               \
        hipLaunchKernelGGL((gemm_tn_bf16_kernel<MODE, NN, KK, NP>), grid, dim3(TB_THREADS), lds_bytes, stream, a, g); \
        rc = check_launch(name);                                                                                    \
    } while (0)
#define TNB_SHAPE(MODE, NP)                                        \
    do {                                                           \
        if (n_q <= 2 && k_h <= 2) TNB_LAUNCH(MODE, 2, 2, NP);      \
        else if (n_q <= 4) TNB_LAUNCH(MODE, 4, 5, NP);             \
        else TNB_LAUNCH(MODE, 5, 5, NP);                           \
    } while (0)
    if (a.amode == A_PLAIN) { if (npass == 3) TNB_SHAPE(A_PLAIN, 3); else TNB_SHAPE(A_PLAIN, 1); }
    else { if (npass == 3) TNB_SHAPE(A_DZ, 3); else TNB_SHAPE(A_DZ, 1); }
#undef TNB_SHAPE
#undef TNB_LAUNCH
    if (rc) return rc;
    const long total = (long)a.N * (a.K + 1);
    TimingScope ts("tn_reduce", stream);
    hipLaunchKernelGGL(tnb_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a.partial, a.splits, a.N, a.K,
                       (long)g.n_tiles * 16, (long)g.k_tiles * 16, a.dW, a.dbias, a.perm);
    return check_launch("tn_reduce");
}

size_t gemm_nt_bf16_wplane_bytes(int N, int K) {
    return (size_t)2 * (cdiv(N, 16) * 16) * (cdiv(K, BF_BK) * BF_BK) * sizeof(__bf16);
}

}  // namespace nrms
