// Tall-skinny GEMM building blocks on the f32-input MFMA (gfx950).
//
// Every dense contraction of the NRMS step has a huge M (tokens: titles*L = 844 800 at the
// bench shape) against tiny N,K in {200,300,900}; weights live in L2, activations stream.
//
//   NT form  C[M,N]  = A'[M,K] . W[N,K]^T          (projections, input gradients)
//   TN form  dW[N,K] += A'[M,N]^T . B'[M,K]        (weight gradients; K gets an extra "ones"
//                                                   column so column K of the result is the
//                                                   bias gradient for free)
// A' / B' are produced by loaders: plain rows, rows gathered from the embedding table by id
// (+ dropout), or the additive-attention dZ = ds*q*(1-T^2) formed on the fly.
#pragma once
#include "common.h"

namespace nrms {

constexpr int NT_BM = 128;            // rows per workgroup tile (4 waves x 32)
constexpr int NT_BK = 32;             // K per LDS stage
constexpr int NT_LS = NT_BK + 4;      // LDS row stride (floats), keeps float4 alignment

enum { A_PLAIN = 0, A_GATHER = 1, A_DZ = 2 };
enum { E_STORE = 0, E_DCTX = 1, E_SCATTER = 2 };

struct NTArgs {
    int M, N, K;
    int rows_per_tile;        // valid rows per workgroup tile (<= NT_BM)
    const float* A; int lda;  // A_PLAIN
    const int64_t* ids;       // A_GATHER / E_SCATTER: row ids
    const float* table;       // A_GATHER: [vocab, K]
    const float* ds;          // A_DZ: [M]
    const float* qv;          // A_DZ: [K]
    const float* T;           // A_DZ: [M, K]
    const float* W;           // [N, K]
    const float* bias;        // [N] or null
    float* C; int ldc;        // output (E_STORE / E_DCTX) or dense table gradient (E_SCATTER, ldc = N)
    const float* wrow;        // E_DCTX: [M] pooling weights
    const float* dout;        // E_DCTX: [n_seq, N]
    int S;                    // E_DCTX: rows per sequence
    Dropout drop;             // A_GATHER: site 0 on A;  E_DCTX: site 1 on C;  E_SCATTER: site 0 on C
};

// ---- NT main loop: acc[mt][nt] covers rows row0 + 32*wave + 16*mt + ..., cols col0 + 16*nt + ...
template <int NT, int AMODE>
__device__ __forceinline__ void gemm_nt_mainloop(const NTArgs& a, int row0, int rows_valid, int col0,
                                                 f32x4 (&acc)[2][NT], float* As, float* Bs) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;

    // each thread stages 4 float4 of A per K step: rows (tid>>3) + 32 i, column quad tid&7
    const int c4 = tid & 7;
    const float* arow[4];
    float ascale[4];
    long agrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 3) + 32 * i;
        const bool ok = r < rows_valid;
        const long g = (long)row0 + r;
        agrow[i] = g;
        arow[i] = nullptr;
        ascale[i] = 0.f;
        if (ok) {
            if (AMODE == A_PLAIN) arow[i] = a.A + g * a.lda;
            else if (AMODE == A_GATHER) arow[i] = a.table + a.ids[g] * (long)a.K;
            else { arow[i] = a.T + g * (long)a.K; ascale[i] = a.ds[g]; }
        }
    }
    constexpr int B_F4 = NT * 16 * 8;                 // float4 per B stage
    constexpr int B_IT = (B_F4 + 255) / 256;

    for (int k0 = 0; k0 < a.K; k0 += NT_BK) {
        const int k = k0 + c4 * 4;
        const bool kok = k < a.K;
        f32x4 av[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (arow[i] != nullptr && kok) {
                v = *reinterpret_cast<const f32x4*>(arow[i] + k);
                if (AMODE == A_GATHER) {
                    if (a.drop.thresh != 0u) {
                        const f32x4 s = dropout_scale4(a.drop.seed, 0u, (uint64_t)(agrow[i] * a.K + k) >> 2,
                                                       a.drop.thresh, a.drop.inv_keep);
                        v *= s;
                    }
                } else if (AMODE == A_DZ) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(a.qv + k);
                    v = ascale[i] * q * (1.0f - v * v);
                }
            }
            av[i] = v;
        }
        f32x4 bv[B_IT];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + 256 * i;
            const int n = col0 + (idx >> 3);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < B_F4 && n < a.N && kok) v = *reinterpret_cast<const f32x4*>(a.W + (long)n * a.K + k);
            bv[i] = v;
        }
        __syncthreads();   // previous stage fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(As + ((tid >> 3) + 32 * i) * NT_LS + c4 * 4) = av[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + 256 * i;
            if (idx < B_F4) *reinterpret_cast<f32x4*>(Bs + (idx >> 3) * NT_LS + c4 * 4) = bv[i];
        }
        __syncthreads();
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            // K permutation: lane quarter kq takes k = 16 ss + 4 kq + e for MFMA e; A and B agree.
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + (32 * wave + r16) * NT_LS + 16 * ss + 4 * kq);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + (32 * wave + 16 + r16) * NT_LS + 16 * ss + 4 * kq);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(Bs + (16 * nt + r16) * NT_LS + 16 * ss + 4 * kq);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][nt] = mfma16(a0[e], b[e], acc[0][nt]);
                    acc[1][nt] = mfma16(a1[e], b[e], acc[1][nt]);
                }
            }
        }
    }
}

int launch_gemm_nt(int amode, int emode, const NTArgs& a, hipStream_t stream, const char* name);

// ---------------------------------------------------------------------------------------
struct TNArgs {
    int M, N, K;              // A' is [M,N]; B' is [M,K] (+ ones column at index K)
    int amode;                // A_PLAIN or A_DZ
    int bmode;                // A_PLAIN or A_GATHER
    const float* A; int lda;  // A_PLAIN
    const float* ds;          // A_DZ: [M]
    const float* qv;          // A_DZ: [N]
    const float* T;           // A_DZ: [M, N]
    const float* B; int ldb;  // plain B
    const int64_t* ids;       // gather B
    const float* table;       // [vocab, K]
    Dropout drop;             // site 0 on gathered B
    float* dW;                // [N, K] accumulated
    float* dbias;             // [N]    accumulated
    float* partial;           // workspace: [splits][n_pad][k_pad]
    int splits;
    int rows_per_split;
};
size_t gemm_tn_workspace_floats(int M, int N, int K, int* splits_out);
int launch_gemm_tn(const TNArgs& a, hipStream_t stream, const char* name);

int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t stream);

}  // namespace nrms
