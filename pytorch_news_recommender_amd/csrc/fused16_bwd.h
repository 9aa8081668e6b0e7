// Shared pieces of the fp16 backward (fused16_bwd.hip: nrms_v0's encoders; fused16_v1_bwd.hip: nrms_v1's news encoder).
#pragma once
#include "fused16.h"

namespace nrms {

constexpr int B16_RED_QKV = 3 * 10 * 32;                  // d(b_qkv) sums, [tile][32]
constexpr int B16_RED = B16_RED_QKV + 2 * F16_QP;         // + d(b_add)[QP] + d(q_vec)[QP]
constexpr int B16_DQ = 96 * 10;                           // dqkv16 row pitch: [head][Q|K|V][32]

struct Bwd16Args {
    int n_seq, S, d, h, dk, q;
    int n_groups;               // ceil(n_seq / F16_WAVES) (+1 with an order list)
    const _Float16* x16;        // [rows][KP]
    const int* pos;             // token -> x16 / dqkv16 row, -1 = padding token; null: row = token
    const int* n_rows;          // device: number of compact rows (with pos); x16 row *n_rows is the padding token's row
    const int64_t* ids;         // non-null: all-padding titles take the closed form
    const int* order;           // [n_cls][n_seq] title lists of launch_title_order (null: titles in index order)
    const int* order_cnt;
    int n_cls;                  // 2: titles with a real token | all-padding titles.  3: long | all-padding | short titles -- then
                                // the pooling kernel hands the attention kernel the d(ctx) of a short title COMPRESSED to its
                                // n + 1 tile rows (the padding tokens' rows summed into one) and the attention kernel puts
                                // two short titles into one 32-row tile, as the forward does (fused16.hip, fused_fwd16p_kernel)
    const _Float16* btiles;     // [4h][32][KP]: the h tiles Wadd_h^T (32 features x QP), then per head W'_q | W_k | W_v
    const float* bqkv32;        // [3h][32]
    const _Float16* qv16;       // [QP]
    const _Float16* ctx16;      // [n_seq*S][DP]   (forward)
    const _Float16* t16;        // [n_seq*S][QP]   (forward)
    const float* w;             // [n_seq*S]       (forward)
    const _Float16* dout16;     // [n_seq][DP]  x loss scale, P16 order
    _Float16* dz16;             // [n_seq*S][QP]
    _Float16* dctx16;           // [n_seq*S][DP]  d(ctx) after the dropout mask, P16 order (written and re-read per wave)
    _Float16* dqkv16;           // [rows][B16_DQ]
    float* red;                 // [gridDim.x][B16_RED] per-workgroup column sums (reduced afterwards, fixed order)
    Dropout drop;
    int v1;                     // pool kernel under fused16_v1_bwd.hip (h, dk = the ten output blocks of W_O): d(ctx)16 is also an
                                // operand of the d(W_O) product over the whole 32-row blocks of the titles with a real token, so
                                // the rows 16..31 of a short title's compressed block are written as zeros
    int dbg;                    // timing experiments only (NRMS_F16_DBG): 4 = no dZ16 / d(ctx)16 stores, 8 = no d(ctx) products
};

// X^T for a 32x32 accumulator X: one product with the k-permuted identity (idf[s][j] = (n == row held as element j))
__device__ __forceinline__ f32x16 transpose32(const f32x16& x, const h8 (&idf)[2]) {
    f32x16 z = mfma32h(acc_frag(x, 0), idf[0], zero16());
    return mfma32h(acc_frag(x, 1), idf[1], z);
}

// sum over the 16 rows a lane holds
__device__ __forceinline__ float regsum(const f32x16& x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) { s0 += x[4 * g]; s1 += x[4 * g + 1]; s2 += x[4 * g + 2]; s3 += x[4 * g + 3]; }
    return (s0 + s1) + (s2 + s3);
}

// X^T from the two operand fragments of X (rows of X = k): one product with the k-permuted identity per fragment
__device__ __forceinline__ f32x16 transpose_frags(const h8& x0, const h8& x1, const h8 (&idf)[2]) {
    f32x16 z = mfma32h(x0, idf[0], zero16());
    return mfma32h(x1, idf[1], z);
}

constexpr int DX_BM = 256, DX_BK = 64, DX_THREADS = 512;
constexpr int DX_SLABS = B16_DQ / DX_BK;                    // 15
constexpr int DX_A_BYTES = DX_BM * DX_BK * 2;               // 32 KB
constexpr int DX_B_BYTES = DX_BK * F16_KP * 2;              // 40 KB
constexpr int DX_SLOT = DX_A_BYTES + DX_B_BYTES;
static_assert(B16_DQ % DX_BK == 0 && F16_KP == 320, "15 slabs of 64; 10 column tiles of 32");
struct Dx16Args {
    int M;                    // upper bound of the rows (sizes the grid)
    const int* m_dev;         // rows actually present (device), or null
    const _Float16* a16;      // [rows][lda]
    const _Float16* xtiles;   // [n_slabs][4][10][2][32][8]
    int lda, n_slabs;         // row pitch of a16 (halves) = 64 n_slabs: B16_DQ / DX_SLABS for nrms_v0
    float* c;                 // [rows][ldc] fp32 (scale removed) -- or OUT16: fp16 [rows][F16_KP], still multiplied by the scale
    int ldc, d;
    const float* sc;          // device: {loss scale, 1 / loss scale}
};


// ---- host wrappers of the kernels both backward paths launch (defined in fused16_bwd.hip)
size_t bwd16_fused_lds(bool two_blocks);                    // dynamic LDS of the pool / attention kernels
int launch_bwd16_pool(const Bwd16Args& a, int n_wg, bool two_blocks, hipStream_t stream);
int launch_dx16(const Dx16Args& g, bool out16, hipStream_t stream);
// absmax (unless fixed_scale > 0) + dout16: sc[0..1] = {scale, 1 / scale}, max_bits = sc + 2
int launch_dout16(long n_seq, int d, int h, int dk, float fixed_scale, float* sc, const float* dout, _Float16* dout16, hipStream_t stream);
// C = A^T B over the token rows, split-M partial slabs, then dW[nmap[n]][kmap[k]] += nscale[n] * sum (kmap -2: dbias[nmap[n]] +=).
// geom 0: row-major operands, 320 x 160 output blocks; 1: fragment-order operands, one 224 x 320 block; 2: fragment order, 320 x 160
int launch_tn16(int geom, const _Float16* A, int lda, int N, const _Float16* B, int ldb, int K, int M, const int* m_dev,
                float* partial, int splits, const int* nmap, const int* kmap, const float* nscale, int ldw, float* dW,
                hipStream_t stream, const char* name, float* dbias = nullptr, const int* blk_list = nullptr, const int* blk_cnt = nullptr,
                int blk_stride = 0);       // blk_*: geom 1 / 2 only, contract over the listed 32-row blocks (long + short title lists)

}  // namespace nrms
