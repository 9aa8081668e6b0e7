// NT / TN GEMM kernels on v_mfma_f32_16x16x4_f32 (see gemm.h for the roles).
#include "gemm.h"

namespace nrms {

// =======================================================================================
// NT: C = A' W^T (+bias) with epilogues
// =======================================================================================
template <int NT, int AMODE, int EMODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (NT_BM + 16 * NT) * NT_BK];

    // 1-D grid, XCD-aware: blocks b, b+8, ... share an XCD (round-robin dispatch); the column
    // tiles of one row tile run back to back there, so the A rows they all stream are fetched
    // into that L2 once (speed only, never correctness).
    const int n_ct = (a.N + NT * 16 - 1) / (NT * 16);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rt = (j / n_ct) * 8 + xcd;
    const int row0 = rt * a.rows_per_tile;
    const int M = a.m_dev != nullptr ? *a.m_dev : a.M;
    if (row0 >= M) return;
    const int rows_valid = min(a.rows_per_tile, M - row0);
    const int col0 = (j % n_ct) * (NT * 16);

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    gemm_nt_mainloop<NT, AMODE>(a, row0, rows_valid, col0, acc, lds);

    // the stage buffers are dead now: reuse them as the per-wave epilogue strips
    static_assert(4 * 8 * (16 * NT + 8) <= 2 * (NT_BM + 16 * NT) * NT_BK, "epilogue strips must fit the stage buffers");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nt_epilogue<NT, EMODE>(a, acc, row0, rows_valid, col0, wave, lane, lds + wave * 8 * (16 * NT + 8));
}

template <int NT, int AMODE, int EMODE>
static int launch_nt_inst(const NTArgs& a, hipStream_t stream, const char* name) {
    dim3 grid(cdiv(cdiv(a.M, a.rows_per_tile), 8) * 8 * cdiv(a.N, NT * 16));
    TimingScope ts(name, stream);
    hipLaunchKernelGGL((gemm_nt_kernel<NT, AMODE, EMODE>), grid, dim3(256), 0, stream, a);
    return check_launch(name);
}

static int pick_nt(int N) {
    // smallest padded width first, then the widest tile (fewer re-reads of A)
    const int cand[5] = {19, 13, 8, 4, 2};
    int best = 19;
    long best_pad = -1;
    for (int i = 0; i < 5; ++i) {
        if (cand[i] < 8 && N > 128) continue;
        const int w = cand[i] * 16;
        const long pad = (long)cdiv(N, w) * w;
        if (best_pad < 0 || pad < best_pad) { best_pad = pad; best = cand[i]; }
    }
    return best;
}

template <int AMODE, int EMODE>
static int launch_nt_mode(const NTArgs& a, hipStream_t stream, const char* name) {
    switch (pick_nt(a.N)) {
        case 19: return launch_nt_inst<19, AMODE, EMODE>(a, stream, name);
        case 13: return launch_nt_inst<13, AMODE, EMODE>(a, stream, name);
        case 8: return launch_nt_inst<8, AMODE, EMODE>(a, stream, name);
        case 4: return launch_nt_inst<4, AMODE, EMODE>(a, stream, name);
        default: return launch_nt_inst<2, AMODE, EMODE>(a, stream, name);
    }
}

int launch_gemm_nt(int amode, int emode, const NTArgs& a, hipStream_t stream, const char* name) {
    if (a.M <= 0) return NRMS_OK;
    if ((a.K & 3) != 0) { set_error("%s: K=%d must be a multiple of 4", name, a.K); return NRMS_EINVAL; }
    if (amode == A_PLAIN && emode == E_STORE) return launch_nt_mode<A_PLAIN, E_STORE>(a, stream, name);
    if (amode == A_DZ && emode == E_DCTX) return launch_nt_mode<A_DZ, E_DCTX>(a, stream, name);
    if (amode == A_DZ && emode == E_STORE) return launch_nt_mode<A_DZ, E_STORE>(a, stream, name);      // segpool.hip: dx = dZ W_add
    set_error("%s: unsupported gemm_nt mode %d/%d", name, amode, emode);
    return NRMS_EINVAL;
}

// =======================================================================================
// TN: dW[N,K(+1)] = sum_m A'[m,N]^T B'[m,K(+ones)] -- split over M, partial slabs + reduce
// =======================================================================================
// Workgroup = 8 waves (4 along N x 2 along K), each wave <= 5x5 tiles of 16x16 (100 accumulator
// registers), so a workgroup owns up to 320 x 160 of the output and walks its slice of M in stages
// of 32 rows.  Software pipeline as in the NT loop: the global loads of stage s+1 are issued
// before the MFMAs of stage s and written to the other LDS buffer after them; one barrier per stage.
constexpr int TN_WN = 4, TN_WK = 2;         // wave grid
constexpr int TN_NTN = 5;                   // max 16-col tiles per wave along N
constexpr int TN_NTK = 5;                   // max 16-col tiles per wave along K
constexpr int TN_MC = 32;                   // rows of M per LDS stage
constexpr int TN_AW = TN_WN * TN_NTN * 16;  // 320 staged A' columns (max)
constexpr int TN_BW = TN_WK * TN_NTK * 16;  // 160 staged B' columns (max)
constexpr int TN_SA = TN_AW + 4;            // 324: stride % 8 == 4 -> conflict-free b32 column reads
constexpr int TN_SB = TN_BW + 4;            // 164
constexpr int TN_STAGE = TN_MC * (TN_SA + TN_SB);          // floats per stage
constexpr int TN_THREADS = 64 * TN_WN * TN_WK;             // 512
constexpr int TN_A_IT = (TN_MC * TN_AW / 4) / TN_THREADS;  // 5 float4 per thread
constexpr int TN_B_IT = (TN_MC * TN_BW / 4 + TN_THREADS - 1) / TN_THREADS;   // 3 (last half-used)

struct TNGeom {
    int n_tiles, k_tiles;     // 16-wide tiles of the padded output
    int n_wg, k_wg;           // workgroup grid over the output
    int n_tpw, k_tpw;         // tiles per workgroup
};

static TNGeom tn_geom(int N, int K) {
    TNGeom g;
    g.n_tiles = cdiv(N, 16);
    g.k_tiles = cdiv(K + 1, 16);            // + ones column
    g.n_wg = cdiv(g.n_tiles, TN_WN * TN_NTN);
    g.k_wg = cdiv(g.k_tiles, TN_WK * TN_NTK);
    // an XCD has 32 CUs and runs whole M-slices (all output tiles of a slice): make the number of
    // output tiles a divisor of 32 so a single round fills every CU
    const int n0 = g.n_wg;
    for (int n = n0; n <= 2 * n0 + 1 && n <= g.n_tiles; ++n)
        if (n * g.k_wg <= 32 && 32 % (n * g.k_wg) == 0) { g.n_wg = n; break; }
    g.n_tpw = cdiv(g.n_tiles, g.n_wg);
    g.k_tpw = cdiv(g.k_tiles, g.k_wg);
    return g;
}

// NTN x NTK = 16x16 tiles per wave, compile-time: every wave issues the same unconditional MFMA
// block (tiles past the matrix edge multiply zero-filled LDS columns and are not stored), so the
// loop body is branch-free -- a VGPR-derived "wave < n" guard makes hipcc wrap each MFMA in an
// exec-mask save/restore + s_waitcnt lgkmcnt(0).
template <int AMODE, int NTN, int NTK>
__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_kernel(TNArgs a, TNGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;

    // XCD-aware decomposition of the 1-D grid: blocks b and b+8 share an XCD, so give every XCD
    // whole M-slices: the n_wg*k_wg output-tile workgroups that stream the SAME rows of A' and B'
    // sit on one L2 (speed only, not correctness).
    const int out_wgs = g.n_wg * g.k_wg;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % out_wgs;
    const int split = (j / out_wgs) * 8 + xcd;
    if (split >= a.splits) return;
    const int bx = tile % g.n_wg, by = tile / g.n_wg;
    const int nt0 = bx * g.n_tpw;                               // first n tile of this WG
    const int nt_cnt = max(0, min(g.n_tpw, g.n_tiles - nt0));
    const int kt0 = by * g.k_tpw;
    const int kt_cnt = max(0, min(g.k_tpw, g.k_tiles - kt0));
    const int n_q = (nt_cnt + TN_WN - 1) / TN_WN, k_h = (kt_cnt + TN_WK - 1) / TN_WK;
    const int my_nt0 = wn * n_q, my_ntn = max(0, min(n_q, nt_cnt - my_nt0));
    const int my_kt0 = wk * k_h, my_ktn = max(0, min(k_h, kt_cnt - my_kt0));

    const int n_cols = nt_cnt * 16, k_cols = kt_cnt * 16;       // staged widths
    const int ncol0 = nt0 * 16, kcol0 = kt0 * 16;
    const int M = a.m_dev != nullptr ? *a.m_dev : a.M;
    const int rps = a.m_dev != nullptr ? cdiv(cdiv(M, a.splits), TN_MC) * TN_MC : a.rows_per_split;
    const int m_begin = split * rps;
    const int m_end = min(M, m_begin + rps);

    // ---- fixed staging slots of this thread.  Loads are unconditional from clamped addresses and keep
    // RAW values; validity and the tanh' factor are applied when the registers go to LDS, a stage later
    // (a predicate or arithmetic next to the load makes hipcc wait for the load right there).
    int a_r[TN_A_IT], a_c[TN_A_IT], a_n[TN_A_IT], b_r[TN_B_IT], b_c[TN_B_IT], b_k[TN_B_IT];
    bool a_ok[TN_A_IT], b_ok[TN_B_IT], b_one[TN_B_IT];
    f32x4 qv_r[TN_A_IT];
#pragma unroll
    for (int i = 0; i < TN_A_IT; ++i) {
        const int sl = tid + TN_THREADS * i;
        a_r[i] = sl / (TN_AW / 4);
        a_c[i] = (sl - a_r[i] * (TN_AW / 4)) * 4;
        a_ok[i] = a_c[i] < n_cols && ncol0 + a_c[i] < a.N;
        a_n[i] = min(ncol0 + a_c[i], a.N - 4);
        if (AMODE == A_DZ) qv_r[i] = *reinterpret_cast<const f32x4*>(a.qv + a_n[i]);
    }
#pragma unroll
    for (int i = 0; i < TN_B_IT; ++i) {
        const int sl = tid + TN_THREADS * i;
        b_r[i] = sl / (TN_BW / 4);
        b_c[i] = (sl - b_r[i] * (TN_BW / 4)) * 4;
        if (b_r[i] >= TN_MC) { b_r[i] = 0; b_c[i] = TN_BW; }     // unused slot
        const int k = kcol0 + b_c[i];
        b_ok[i] = b_c[i] < k_cols && k < a.K;
        b_one[i] = b_c[i] < k_cols && k == a.K;                   // ones column -> bias gradient
        b_k[i] = min(k, a.K - 4);
    }

    f32x4 av[TN_A_IT], bv[TN_B_IT];
    float dsv[TN_A_IT];
    auto load_stage = [&](int m0) {
#pragma unroll
        for (int i = 0; i < TN_A_IT; ++i) {
            const long m = min(m0 + a_r[i], m_end - 1);
            if (AMODE == A_PLAIN) {
                av[i] = *reinterpret_cast<const f32x4*>(a.A + m * a.lda + a_n[i]);
            } else {
                av[i] = *reinterpret_cast<const f32x4*>(a.T + m * (long)a.N + a_n[i]);
                dsv[i] = a.ds[m];
            }
        }
#pragma unroll
        for (int i = 0; i < TN_B_IT; ++i) {
            const long m = min(m0 + b_r[i], m_end - 1);
            bv[i] = *reinterpret_cast<const f32x4*>(a.B + m * a.ldb + b_k[i]);
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto store_stage = [&](int m0, float* st) {
#pragma unroll
        for (int i = 0; i < TN_A_IT; ++i) {
            f32x4 v = av[i];
            if (AMODE == A_DZ) v = dsv[i] * qv_r[i] * (1.0f - v * v);
            *reinterpret_cast<f32x4*>(st + a_r[i] * TN_SA + a_c[i]) = (a_ok[i] && m0 + a_r[i] < m_end) ? v : zero4;
        }
#pragma unroll
        for (int i = 0; i < TN_B_IT; ++i) {
            if (b_c[i] >= TN_BW) continue;
            const bool mok = m0 + b_r[i] < m_end;
            f32x4 v = (b_ok[i] && mok) ? bv[i] : zero4;
            if (b_one[i] && mok) v[0] = 1.0f;
            *reinterpret_cast<f32x4*>(st + TN_MC * TN_SA + b_r[i] * TN_SB + b_c[i]) = v;
        }
    };

    f32x4 acc[NTN][NTK];
#pragma unroll
    for (int i = 0; i < NTN; ++i)
#pragma unroll
        for (int jj = 0; jj < NTK; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int n_stage = (m_end - m_begin + TN_MC - 1) / TN_MC;
    if (n_stage > 0) {                 // an empty split (compacted M) still writes its all-zero slab below
    load_stage(m_begin);
    store_stage(m_begin, lds);
    __syncthreads();
    }
    for (int s = 0; s < n_stage; ++s) {
        const float* As = lds + (s & 1) * TN_STAGE;
        const float* Bs = As + TN_MC * TN_SA;
        // branch-free: past the split's end the loads re-read its last row and are zeroed at the store
        load_stage(m_begin + (s + 1) * TN_MC);
        // 32 rows of M = 2 sub-chunks x 4 MFMA k-steps (m = 16 sub + 4 kq + e)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mrow = 16 * sub + 4 * kq + e;
                float bf[NTK], af[NTN];
#pragma unroll
                for (int jj = 0; jj < NTK; ++jj) bf[jj] = Bs[mrow * TN_SB + (my_kt0 + jj) * 16 + r16];
#pragma unroll
                for (int i = 0; i < NTN; ++i) af[i] = As[mrow * TN_SA + (my_nt0 + i) * 16 + r16];
#pragma unroll
                for (int i = 0; i < NTN; ++i)
#pragma unroll
                    for (int jj = 0; jj < NTK; ++jj) acc[i][jj] = mfma16(af[i], bf[jj], acc[i][jj]);
            }
        }
        store_stage(m_begin + (s + 1) * TN_MC, lds + ((s + 1) & 1) * TN_STAGE);
        __syncthreads();
    }
    // ---- partial slab: [split][n_pad][k_pad]
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

__global__ void tn_reduce_kernel(const float* partial, int splits, int N, int K, long n_pad, long k_pad,
                                 float* dW, float* dbias, HeadPerm perm) {
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

static int tn_splits(int M, const TNGeom& g) {
    // one 8-wave workgroup per CU, one round: each of the 8 XCDs (32 CUs) runs 32/out_wgs slices
    const int out_wgs = g.n_wg * g.k_wg;
    int splits = 8 * (out_wgs <= 32 ? 32 / out_wgs : 1);
    const int max_splits = cdiv(M, 8 * TN_MC);       // at least 256 rows per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return splits;
}

size_t gemm_tn_workspace_floats(int M, int N, int K, int* splits_out) {
    const TNGeom g = tn_geom(N, K);
    const int splits = tn_splits(M, g);
    if (splits_out) *splits_out = splits;
    return (size_t)splits * g.n_tiles * 16 * g.k_tiles * 16;
}

int launch_gemm_tn(const TNArgs& a_in, hipStream_t stream, const char* name) {
    if (a_in.M <= 0) return NRMS_OK;
    TNArgs a = a_in;
    const TNGeom g = tn_geom(a.N, a.K);
    if ((a.N & 3) != 0 || (a.K & 3) != 0) { set_error("%s: N,K must be multiples of 4", name); return NRMS_EINVAL; }
    a.splits = tn_splits(a.M, g);
    a.rows_per_split = cdiv(cdiv(a.M, a.splits), TN_MC) * TN_MC;
    a.splits = cdiv(a.M, a.rows_per_split);
    dim3 grid(cdiv(a.splits, 8) * 8 * g.n_wg * g.k_wg);
    constexpr size_t lds_bytes = 2 * TN_STAGE * sizeof(float);
    const int n_q = cdiv(g.n_tpw, TN_WN), k_h = cdiv(g.k_tpw, TN_WK);     // tiles per wave actually needed
    int rc = NRMS_OK;
#define TN_LAUNCH(MODE, NN, KK)                                                                                     \
    do {                                                                                                            \
        const void* fn = (const void*)gemm_tn_kernel<MODE, NN, KK>;                                                 \
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);   \
        if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return NRMS_ELAUNCH; } \
        TimingScope ts(name, stream);                                                                               \
        hipLaunchKernelGGL((gemm_tn_kernel<MODE, NN, KK>), grid, dim3(TN_THREADS), lds_bytes, stream, a, g);        \
        rc = check_launch(name);                                                                                    \
    } while (0)
    if (a.amode == A_PLAIN) {
        if (n_q <= 2 && k_h <= 2) TN_LAUNCH(A_PLAIN, 2, 2);
        else if (n_q <= 4) TN_LAUNCH(A_PLAIN, 4, 5);
        else TN_LAUNCH(A_PLAIN, 5, 5);
    } else {
        if (n_q <= 2 && k_h <= 2) TN_LAUNCH(A_DZ, 2, 2);
        else if (n_q <= 4) TN_LAUNCH(A_DZ, 4, 5);
        else TN_LAUNCH(A_DZ, 5, 5);
    }
#undef TN_LAUNCH
    if (rc) return rc;
    const long total = (long)a.N * (a.K + 1);
    TimingScope ts("tn_reduce", stream);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a.partial, a.splits, a.N, a.K,
                       (long)g.n_tiles * 16, (long)g.k_tiles * 16, a.dW, a.dbias, a.perm);
    return check_launch("tn_reduce");
}

// =======================================================================================
__global__ void transpose_kernel(const float* in, float* out, int rows, int cols, HeadPerm perm) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(long)perm.src(r) * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = bx + j, r = by + threadIdx.x;     // out[c][r]
        if (c < cols && r < rows) out[(long)c * rows + r] = tile[threadIdx.x][j];
    }
}

int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t stream, HeadPerm perm) {
    TimingScope ts("transpose", stream);
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, stream, in, out, rows,
                       cols, perm);
    return check_launch("transpose");
}

__global__ void permute_rows_kernel(const float* w, const float* b, float* w2, float* b2, int rows, int cols,
                                    HeadPerm perm) {
    const int r = blockIdx.x, rs = perm.src(r);
    for (int c = threadIdx.x * 4; c < cols; c += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(w2 + (long)r * cols + c) = *reinterpret_cast<const f32x4*>(w + (long)rs * cols + c);
    if (threadIdx.x == 0) b2[r] = b[rs];
}

int launch_permute_rows(const float* w, const float* b, float* w2, float* b2, int rows, int cols, HeadPerm perm,
                        hipStream_t stream) {
    if ((cols & 3) != 0) { set_error("permute_rows: cols=%d must be a multiple of 4", cols); return NRMS_EINVAL; }
    TimingScope ts("permute_rows", stream);
    hipLaunchKernelGGL(permute_rows_kernel, dim3(rows), dim3(128), 0, stream, w, b, w2, b2, rows, cols, perm);
    return check_launch("permute_rows");
}

}  // namespace nrms
