// NT / TN GEMM kernels on v_mfma_f32_16x16x4_f32 (see gemm.h for the roles).
#include "gemm.h"

namespace nrms {

// =======================================================================================
// NT: C = A' W^T (+bias) with epilogues
// =======================================================================================
template <int NT, int AMODE, int EMODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTArgs a) {
    __shared__ __attribute__((aligned(16))) float As[NT_BM * NT_LS];
    __shared__ __attribute__((aligned(16))) float Bs[NT * 16 * NT_LS];

    const int row0 = blockIdx.x * a.rows_per_tile;
    const int rows_valid = min(a.rows_per_tile, a.M - row0);
    const int col0 = blockIdx.y * (NT * 16);

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    gemm_nt_mainloop<NT, AMODE>(a, row0, rows_valid, col0, acc, As, Bs);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rl = 32 * wave + 16 * mt + 4 * kq + reg;
            if (rl >= rows_valid) continue;
            const long g = (long)row0 + rl;
            long dst_row = g;
            float wr = 0.f;
            const float* dout_row = nullptr;
            if (EMODE == E_SCATTER) {
                dst_row = a.ids[g];
                if (dst_row == 0) continue;          // padding_idx = 0: no gradient to the pad row
            }
            if (EMODE == E_DCTX) {
                wr = a.wrow[g];
                dout_row = a.dout + (g / a.S) * (long)a.N;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = col0 + 16 * nt + r16;
                if (n >= a.N) continue;
                float v = acc[mt][nt][reg];
                if (EMODE == E_STORE) {
                    if (a.bias != nullptr) v += a.bias[n];
                    a.C[dst_row * a.ldc + n] = v;
                } else if (EMODE == E_DCTX) {
                    v += wr * dout_row[n];
                    if (a.drop.thresh != 0u)
                        v *= dropout_scale1(a.drop.seed, 1u, (uint64_t)(g * a.N + n), a.drop.thresh, a.drop.inv_keep);
                    a.C[g * a.ldc + n] = v;
                } else {  // E_SCATTER
                    if (a.drop.thresh != 0u)
                        v *= dropout_scale1(a.drop.seed, 0u, (uint64_t)(g * a.N + n), a.drop.thresh, a.drop.inv_keep);
                    atomicAdd(a.C + dst_row * a.ldc + n, v);
                }
            }
        }
    }
}

template <int NT, int AMODE, int EMODE>
static int launch_nt_inst(const NTArgs& a, hipStream_t stream, const char* name) {
    dim3 grid(cdiv(a.M, a.rows_per_tile), cdiv(a.N, NT * 16));
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
    if (amode == A_GATHER && emode == E_STORE) return launch_nt_mode<A_GATHER, E_STORE>(a, stream, name);
    if (amode == A_PLAIN && emode == E_STORE) return launch_nt_mode<A_PLAIN, E_STORE>(a, stream, name);
    if (amode == A_PLAIN && emode == E_SCATTER) return launch_nt_mode<A_PLAIN, E_SCATTER>(a, stream, name);
    if (amode == A_DZ && emode == E_DCTX) return launch_nt_mode<A_DZ, E_DCTX>(a, stream, name);
    set_error("%s: unsupported gemm_nt mode %d/%d", name, amode, emode);
    return NRMS_EINVAL;
}

// =======================================================================================
// TN: dW[N,K(+1)] = sum_m A'[m,N]^T B'[m,K(+ones)] -- split over M, partial slabs + reduce
// =======================================================================================
constexpr int TN_NTN = 10;                  // max 16-col tiles per wave along N  (2 waves)
constexpr int TN_NTK = 5;                   // max 16-col tiles per wave along K  (2 waves)
constexpr int TN_MC = 32;                   // rows of M per LDS stage
constexpr int TN_SA = 2 * TN_NTN * 16 + 4;  // 324: stride % 8 == 4 -> conflict-free b32 column reads
constexpr int TN_SB = 2 * TN_NTK * 16 + 4;  // 164

struct TNGeom {
    int n_tiles, k_tiles;     // 16-wide tiles of the padded output
    int n_wg, k_wg;           // workgroup grid over the output
    int n_tpw, k_tpw;         // tiles per workgroup
};

static TNGeom tn_geom(int N, int K) {
    TNGeom g;
    g.n_tiles = cdiv(N, 16);
    g.k_tiles = cdiv(K + 1, 16);            // + ones column
    g.n_wg = cdiv(g.n_tiles, 2 * TN_NTN);
    g.k_wg = cdiv(g.k_tiles, 2 * TN_NTK);
    g.n_tpw = cdiv(g.n_tiles, g.n_wg);
    g.k_tpw = cdiv(g.k_tiles, g.k_wg);
    return g;
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TNArgs a, TNGeom g) {
    __shared__ __attribute__((aligned(16))) float As[TN_MC * TN_SA];
    __shared__ __attribute__((aligned(16))) float Bs[TN_MC * TN_SB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;

    const int nt0 = blockIdx.x * g.n_tpw;                       // first n tile of this WG
    const int nt_cnt = min(g.n_tpw, g.n_tiles - nt0);
    const int kt0 = blockIdx.y * g.k_tpw;
    const int kt_cnt = min(g.k_tpw, g.k_tiles - kt0);
    const int n_half = (nt_cnt + 1) >> 1, k_half = (kt_cnt + 1) >> 1;
    const int my_nt0 = wn * n_half, my_ntn = wn == 0 ? n_half : nt_cnt - n_half;
    const int my_kt0 = wk * k_half, my_ktn = wk == 0 ? k_half : kt_cnt - k_half;

    const int n_cols = nt_cnt * 16, k_cols = kt_cnt * 16;       // staged widths
    const int ncol0 = nt0 * 16, kcol0 = kt0 * 16;
    const int m_begin = blockIdx.z * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);

    f32x4 acc[TN_NTN][TN_NTK];
#pragma unroll
    for (int i = 0; i < TN_NTN; ++i)
#pragma unroll
        for (int j = 0; j < TN_NTK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_f4_per_row = n_cols >> 2, b_f4_per_row = k_cols >> 2;
    for (int m0 = m_begin; m0 < m_end; m0 += TN_MC) {
        __syncthreads();
        // ---- stage A' chunk [32][n_cols]
        for (int idx = tid; idx < TN_MC * a_f4_per_row; idx += 256) {
            const int r = idx / a_f4_per_row, c = (idx - r * a_f4_per_row) * 4;
            const long m = (long)m0 + r;
            const int n = ncol0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end && n < a.N) {
                if (AMODE == A_PLAIN) {
                    v = *reinterpret_cast<const f32x4*>(a.A + m * a.lda + n);
                } else {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(a.T + m * (long)a.N + n);
                    const f32x4 q = *reinterpret_cast<const f32x4*>(a.qv + n);
                    v = a.ds[m] * q * (1.0f - t * t);
                }
            }
            *reinterpret_cast<f32x4*>(As + r * TN_SA + c) = v;
        }
        // ---- stage B' chunk [32][k_cols] (column K = 1.0 -> bias gradient)
        for (int idx = tid; idx < TN_MC * b_f4_per_row; idx += 256) {
            const int r = idx / b_f4_per_row, c = (idx - r * b_f4_per_row) * 4;
            const long m = (long)m0 + r;
            const int k = kcol0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
                if (k < a.K) {
                    if (BMODE == A_PLAIN) {
                        v = *reinterpret_cast<const f32x4*>(a.B + m * a.ldb + k);
                    } else {
                        v = *reinterpret_cast<const f32x4*>(a.table + a.ids[m] * (long)a.K + k);
                        if (a.drop.thresh != 0u)
                            v *= dropout_scale4(a.drop.seed, 0u, (uint64_t)(m * a.K + k) >> 2, a.drop.thresh,
                                                a.drop.inv_keep);
                    }
                } else if (k == a.K) {
                    v[0] = 1.0f;
                }
            }
            *reinterpret_cast<f32x4*>(Bs + r * TN_SB + c) = v;
        }
        __syncthreads();
        // ---- 32 rows of M = 2 sub-chunks x 4 MFMA k-steps (m = 16 sub + 4 kq + e)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mrow = 16 * sub + 4 * kq + e;
                float bf[TN_NTK];
#pragma unroll
                for (int j = 0; j < TN_NTK; ++j)
                    bf[j] = j < my_ktn ? Bs[mrow * TN_SB + (my_kt0 + j) * 16 + r16] : 0.f;
#pragma unroll
                for (int i = 0; i < TN_NTN; ++i) {
                    if (i < my_ntn) {
                        const float af = As[mrow * TN_SA + (my_nt0 + i) * 16 + r16];
#pragma unroll
                        for (int j = 0; j < TN_NTK; ++j)
                            if (j < my_ktn) acc[i][j] = mfma16(af, bf[j], acc[i][j]);
                    }
                }
            }
        }
    }
    // ---- partial slab: [split][n_pad][k_pad]
    const long n_pad = (long)g.n_tiles * 16, k_pad = (long)g.k_tiles * 16;
    float* slab = a.partial + (long)blockIdx.z * n_pad * k_pad;
#pragma unroll
    for (int i = 0; i < TN_NTN; ++i) {
        if (i >= my_ntn) continue;
#pragma unroll
        for (int j = 0; j < TN_NTK; ++j) {
            if (j >= my_ktn) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long n = ncol0 + (my_nt0 + i) * 16 + 4 * kq + reg;
                const long k = kcol0 + (my_kt0 + j) * 16 + r16;
                slab[n * k_pad + k] = acc[i][j][reg];
            }
        }
    }
}

__global__ void tn_reduce_kernel(const float* partial, int splits, int N, int K, long n_pad, long k_pad,
                                 float* dW, float* dbias) {
    const long total = (long)N * (K + 1);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx / (K + 1), k = idx - n * (K + 1);
        float s = 0.f;
        for (int sp = 0; sp < splits; ++sp) s += partial[(long)sp * n_pad * k_pad + n * k_pad + k];
        if (k < K) dW[n * K + k] += s;
        else if (dbias != nullptr) dbias[n] += s;
    }
}

static int tn_splits(int M, const TNGeom& g) {
    const int out_wgs = g.n_wg * g.k_wg;
    int splits = cdiv(768, out_wgs);                 // ~3 workgroups per CU in flight
    const int max_splits = cdiv(M, 4 * TN_MC);       // at least 128 rows per split
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
    dim3 grid(g.n_wg, g.k_wg, a.splits);
    {
        TimingScope ts(name, stream);
        if (a.amode == A_PLAIN && a.bmode == A_PLAIN)
            hipLaunchKernelGGL((gemm_tn_kernel<A_PLAIN, A_PLAIN>), grid, dim3(256), 0, stream, a, g);
        else if (a.amode == A_PLAIN && a.bmode == A_GATHER)
            hipLaunchKernelGGL((gemm_tn_kernel<A_PLAIN, A_GATHER>), grid, dim3(256), 0, stream, a, g);
        else if (a.amode == A_DZ && a.bmode == A_PLAIN)
            hipLaunchKernelGGL((gemm_tn_kernel<A_DZ, A_PLAIN>), grid, dim3(256), 0, stream, a, g);
        else { set_error("%s: unsupported gemm_tn mode", name); return NRMS_EINVAL; }
        int rc = check_launch(name);
        if (rc) return rc;
    }
    const long total = (long)a.N * (a.K + 1);
    TimingScope ts("tn_reduce", stream);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a.partial, a.splits, a.N, a.K,
                       (long)g.n_tiles * 16, (long)g.k_tiles * 16, a.dW, a.dbias);
    return check_launch("tn_reduce");
}

// =======================================================================================
__global__ void transpose_kernel(const float* in, float* out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = bx + j, r = by + threadIdx.x;     // out[c][r]
        if (c < cols && r < rows) out[(long)c * rows + r] = tile[threadIdx.x][j];
    }
}

int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t stream) {
    TimingScope ts("transpose", stream);
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, stream, in, out, rows,
                       cols);
    return check_launch("transpose");
}

}  // namespace nrms
