// Tall-skinny GEMM building blocks on the f32-input MFMA (gfx950).
//
// Every dense contraction of the NRMS step has a huge M (tokens: titles*L = 844 800 at the
// bench shape) against tiny N,K in {200,300,900}; weights live in L2, activations stream.
//
//   NT form  C[M,N]  = A'[M,K] . W[N,K]^T          (projections, input gradients)
//   TN form  dW[N,K] += A'[M,N]^T . B'[M,K]        (weight gradients; K gets an extra "ones"
//                                                   column so column K of the result is the
//                                                   bias gradient for free)
// The GEMM waves carry NO gather / RNG work: measured on MI355X, VALU work inside an MFMA wave
// comes straight out of MFMA time (the Philox-dropout gather loader ran the matrix pipe 58 %
// busy against 75 % for a plain loader), so embedding gather+dropout and the dropout+scatter of
// the embedding gradient are separate HBM-bound kernels (embed.hip).  The one fused transform
// left is the additive-attention dZ = ds*q*(1-T^2) (three VALU ops per element).
#pragma once
#include "common.h"

namespace nrms {

constexpr int NT_BM = 128;            // rows per workgroup tile (4 waves x 32)
constexpr int NT_BK = 16;             // K per LDS stage (one 64-byte row segment)
constexpr int NT_STAGE_A = NT_BM * NT_BK;                 // floats per A stage

enum { A_PLAIN = 0, A_DZ = 2 };
enum { E_STORE = 0, E_DCTX = 1 };

struct NTArgs {
    int M, N, K;
    int rows_per_tile;        // valid rows per workgroup tile (<= NT_BM)
    const float* A; int lda;  // A_PLAIN
    const float* ds;          // A_DZ: [M]
    const float* qv;          // A_DZ: [K]
    const float* T;           // A_DZ: [M, K]
    const float* W;           // [N, K]
    const float* bias;        // [N] or null
    float* C; int ldc;        // output
    const float* wrow;        // E_DCTX: [M] pooling weights
    const float* dout;        // E_DCTX: [n_seq, N]
    int S;                    // E_DCTX: rows per sequence
    Dropout drop;             // optional epilogue dropout (thresh != 0), Philox site 1, element index
                              // g*N + n -- only the output-projection topology (nrms_v1) needs it here
    // Row compaction (A_PLAIN only; both null = dense): row r of the product is row a_rows[r] of A, and
    // the number of rows is read from device memory (*m_dev <= M; M sizes the grid).  C rows stay compact.
    const int* a_rows;
    const int* m_dev;
    const int* c_rows;        // optional: row r of the product is stored to row c_rows[r] of C
};

// LDS image of a K-stage: [rows][4 chunks of 16 B], the chunk index XOR-swizzled per 4-row group
// so that the ds_read_b128 fragment reads (lane = 16*kq + row, one chunk each) hit 16 distinct
// 16-byte slots in every hardware lane group {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
__device__ __forceinline__ int nt_swz(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

// ---- NT main loop: acc[mt][nt] covers rows row0 + 32*wave + 16*mt + ..., cols col0 + 16*nt + ...
// Software pipeline: global loads of stage s+1 are issued before the MFMAs of stage s and written
// to the other LDS buffer after them; one barrier per stage.
//   lds: 2 * (NT_BM + 16*NT) * NT_BK floats
template <int NT, int AMODE>
__device__ __forceinline__ void gemm_nt_mainloop(const NTArgs& a, int row0, int rows_valid, int col0,
                                                 f32x4 (&acc)[2][NT], float* lds) {
    constexpr int STAGE = (NT_BM + 16 * NT) * NT_BK;     // floats per stage (A then B)
    constexpr int B_F4 = NT * 16 * 4;                    // float4 per B stage
    constexpr int B_IT = (B_F4 + 255) / 256;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;

    // ---- staging assignment: A rows (tid>>2) and (tid>>2)+64, 16-byte chunk tid&3
    const int chunk = tid & 3;
    // Every load is unconditional from a clamped address and keeps the RAW value; row / column / K-tail
    // validity and the tanh' factor are applied when the registers are written to LDS, a stage later
    // (see gemm_bf16.hip: a predicate or arithmetic next to the load drags the wait up to it).
    const float* arow[2];
    float ascale[2];
    bool a_ok[2];
    int a_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 2) + 64 * i;
        const long g = (long)row0 + min(r, rows_valid - 1);
        a_ok[i] = r < rows_valid;
        a_lds[i] = (r * 4 + (chunk ^ nt_swz(r))) * 4;
        if (AMODE == A_PLAIN) { arow[i] = a.A + (a.a_rows != nullptr ? (long)a.a_rows[g] : g) * a.lda; ascale[i] = 0.f; }
        else { arow[i] = a.T + g * (long)a.K; ascale[i] = a.ds[g]; }
    }
    const float* brow[B_IT];
    bool b_ok[B_IT];
    int b_lds[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 2;
        const int n = col0 + r;
        b_ok[i] = idx < B_F4 && n < a.N;
        brow[i] = a.W + (long)min(n, a.N - 1) * a.K;
        b_lds[i] = idx < B_F4 ? NT_STAGE_A + (r * 4 + (chunk ^ nt_swz(r))) * 4 : -1;
    }
    // fragment read offsets (floats) inside a stage
    const int fsw = kq ^ nt_swz(r16);
    const int a_frag0 = ((32 * wave + r16) * 4 + fsw) * 4;
    const int a_frag1 = ((32 * wave + 16 + r16) * 4 + fsw) * 4;
    const int b_frag = NT_STAGE_A + (r16 * 4 + fsw) * 4;

    f32x4 av[2], bv[B_IT], qv4;
    const int k_last = a.K - 4;                  // K % 4 == 0 (checked by the launcher)
    auto load_stage = [&](int k0) {
        const int k = min(k0 + chunk * 4, k_last);
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const f32x4*>(arow[i] + k);
        if (AMODE == A_DZ) qv4 = *reinterpret_cast<const f32x4*>(a.qv + k);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) bv[i] = *reinterpret_cast<const f32x4*>(brow[i] + k);
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto store_stage = [&](int k0, float* st) {
        const bool kok = k0 + chunk * 4 < a.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v = av[i];
            if (AMODE == A_DZ) v = ascale[i] * qv4 * (1.0f - v * v);
            *reinterpret_cast<f32x4*>(st + a_lds[i]) = (a_ok[i] && kok) ? v : zero4;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (b_lds[i] >= 0) *reinterpret_cast<f32x4*>(st + b_lds[i]) = (b_ok[i] && kok) ? bv[i] : zero4;
    };

    const int n_stage = (a.K + NT_BK - 1) / NT_BK;
    load_stage(0);
    store_stage(0, lds);
    __syncthreads();
    for (int s = 0; s < n_stage; ++s) {
        const float* cur = lds + (s & 1) * STAGE;
        load_stage((s + 1) * NT_BK);             // branch-free: past K the address is clamped, the store writes zeros
        // K permutation: lane quarter kq takes k = 4 kq + e for MFMA e (A and B agree)
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(cur + a_frag0);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(cur + a_frag1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(cur + b_frag + nt * 16 * NT_BK);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][nt] = mfma16(a0[e], b[e], acc[0][nt]);
                acc[1][nt] = mfma16(a1[e], b[e], acc[1][nt]);
            }
        }
        store_stage((s + 1) * NT_BK, lds + ((s + 1) & 1) * STAGE);
        __syncthreads();
    }
}

// ---- NT epilogue: accumulator tile -> HBM in whole rows.
// The MFMA C layout puts one column per lane (16 lanes = one 64-byte row fragment), so storing from
// registers costs one scalar store instruction per accumulator register and touches four partial
// lines each: store-issue-bound (it capped the bf16 kernels at 21 % MFMA utilisation).  Instead each
// wave bounces 8 rows at a time through a private LDS strip [8][16 NT + 8] (stride = 8 mod 16 floats:
// conflict-free ds_write_b32) and writes float4 per lane = contiguous 1 KiB per wave-instruction,
// applying bias / w_s*dout / dropout (ONE Philox call per float4) on the way out.
//   wave_lds: 8 * (16 NT + 8) floats private to the calling wave.
template <int NT, int EMODE>
__device__ __forceinline__ void nt_epilogue(const NTArgs& g, const f32x4 (&acc)[2][NT], int row0, int rows_valid,
                                            int col0, int wave, int lane, float* wave_lds) {
    // One store instruction = one whole row (or 64 / LPR rows of a narrow tile): with a row per
    // instruction the row index is wave-uniform, so the C address is scalar arithmetic, and a lane's
    // column -- hence its bias slice and its validity -- is fixed for the whole tile.  (The first version
    // walked a flat index over the strip: a division, a 64-bit mad and a bias load per float4, ~2.6k
    // VALU instructions per wave and tile, a fifth of the QKV GEMM.)
    constexpr int S = 16 * NT + 8;
    constexpr int F4_ROW = 4 * NT;               // float4 per row of the tile
    constexpr int LPR = F4_ROW >= 64 ? 64 : F4_ROW;
    constexpr int RPI = 64 / LPR;                // rows per store instruction: 1, 2 (NT = 8) or 4 (NT = 4)
    constexpr int REM = F4_ROW - LPR * (RPI == 1 ? 1 : 0) > 0 && F4_ROW > 64 ? F4_ROW - 64 : 0;
    static_assert(8 % RPI == 0, "strip rows must divide into store instructions");
    const int r16 = lane & 15, kq = lane >> 4;
    const int sub = RPI == 1 ? 0 : lane / LPR;   // which of the RPI rows this lane stores
    const int c4a = RPI == 1 ? lane : lane - sub * LPR;
    const int na = col0 + 4 * c4a;
    const bool ok_a = c4a < F4_ROW && sub < RPI && na < g.N;
    f32x4 bias_a = {0.f, 0.f, 0.f, 0.f};
    if (EMODE == E_STORE && g.bias != nullptr && ok_a) bias_a = *reinterpret_cast<const f32x4*>(g.bias + na);
    const float inv_S = EMODE == E_STORE ? 0.f : 1.0f / (float)g.S;

    // bias / w_s * dout of one float4 of row gr at column n
    auto fix = [&](f32x4 v, long gr, int n, const f32x4& bias4) {
        if (EMODE == E_STORE) return v + bias4;
        const long seq = (long)(((float)gr + 0.5f) * inv_S);               // exact for gr < 2^22
        return v + g.wrow[gr] * *reinterpret_cast<const f32x4*>(g.dout + seq * (long)g.N + n);
    };
    const bool dropped = g.drop.thresh != 0u;                              // uniform
    auto rem_pos = [&](int idx, int rbase, int& sr, int& c4, int& rl, int& n) {
        constexpr int REMD = REM > 0 ? REM : 1;
        sr = idx / REMD; c4 = 64 + idx - sr * REMD;
        rl = rbase + 4 * (sr >> 1) + (sr & 1);
        n = col0 + 4 * c4;
        return rl < rows_valid && n < g.N;
    };

#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // rows 4 kq + 2 h + b of the 16-row tile -> strip row 2 kq + b
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wave_lds[(2 * kq + b) * S + 16 * nt + r16] = acc[mt][nt][2 * h + b];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int rbase = 32 * wave + 16 * mt + 2 * h;                   // wave-uniform
            if (dropped) {
                // dropout pass, in place in the strip (each lane rewrites exactly the float4s it stores
                // below): kept out of the unrolled store pass so the Philox code exists once per strip
#pragma unroll 1
                for (int it = 0; it < 8 / RPI; ++it) {
                    const int sr = it * RPI + sub;
                    const int rl = rbase + 4 * (sr >> 1) + (sr & 1);
                    if (ok_a && rl < rows_valid) {
                        const long gr = (long)row0 + rl;
                        f32x4* q = reinterpret_cast<f32x4*>(wave_lds + sr * S + 4 * c4a);
                        *q = fix(*q, gr, na, bias_a) *
                             dropout_scale4(g.drop.seed, 1u, (uint64_t)(gr * g.N + na) >> 2, g.drop.thresh, g.drop.inv_keep);
                    }
                }
                if (REM > 0) {
                    for (int idx = lane; idx < 8 * REM; idx += 64) {
                        int sr, c4, rl, n;
                        if (!rem_pos(idx, rbase, sr, c4, rl, n)) continue;
                        const long gr = (long)row0 + rl;
                        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                        if (EMODE == E_STORE && g.bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
                        f32x4* q = reinterpret_cast<f32x4*>(wave_lds + sr * S + 4 * c4);
                        *q = fix(*q, gr, n, b4) *
                             dropout_scale4(g.drop.seed, 1u, (uint64_t)(gr * g.N + n) >> 2, g.drop.thresh, g.drop.inv_keep);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // store pass: unrolled, so the 8 strip reads (and the dout loads) are all in flight together
#pragma unroll
            for (int it = 0; it < 8 / RPI; ++it) {
                const int sr = it * RPI + sub;                               // compile-time when RPI == 1
                const int rl = rbase + 4 * (sr >> 1) + (sr & 1);
                if (ok_a && rl < rows_valid) {
                    const long gr = (long)row0 + rl;
                    f32x4 v = *reinterpret_cast<const f32x4*>(wave_lds + sr * S + 4 * c4a);
                    if (!dropped) v = fix(v, gr, na, bias_a);
                    *reinterpret_cast<f32x4*>(g.C + (g.c_rows != nullptr ? (long)g.c_rows[gr] : gr) * g.ldc + na) = v;
                }
            }
            if (REM > 0) {
                // columns 64.. of the wide tile: REM float4 per row, 8 rows, packed densely over the lanes
                for (int idx = lane; idx < 8 * REM; idx += 64) {
                    int sr, c4, rl, n;
                    if (!rem_pos(idx, rbase, sr, c4, rl, n)) continue;
                    const long gr = (long)row0 + rl;
                    f32x4 v = *reinterpret_cast<const f32x4*>(wave_lds + sr * S + 4 * c4);
                    if (!dropped) {
                        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                        if (EMODE == E_STORE && g.bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
                        v = fix(v, gr, n, b4);
                    }
                    *reinterpret_cast<f32x4*>(g.C + (g.c_rows != nullptr ? (long)g.c_rows[gr] : gr) * g.ldc + n) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

int launch_gemm_nt(int amode, int emode, const NTArgs& a, hipStream_t stream, const char* name);

// ---------------------------------------------------------------------------------------
struct TNArgs {
    int M, N, K;              // A' is [M,N]; B' is [M,K] (+ ones column at index K)
    int amode;                // A_PLAIN or A_DZ
    const float* A; int lda;  // A_PLAIN
    const float* ds;          // A_DZ: [M]
    const float* qv;          // A_DZ: [N]
    const float* T;           // A_DZ: [M, N]
    const float* B; int ldb;  // [M, K]
    float* dW;                // [N, K] accumulated
    float* dbias;             // [N]    accumulated
    float* partial;           // workspace: [splits][n_pad][k_pad]
    int splits;
    int rows_per_split;
    HeadPerm perm;            // row n' of the product is accumulated into dW / dbias row perm.src(n')
    // Compacted operands: the sum runs over m < *m_dev (<= M, which sizes the grid and the slabs); null = M
    const int* m_dev;
};
size_t gemm_tn_workspace_floats(int M, int N, int K, int* splits_out);
int launch_gemm_tn(const TNArgs& a, hipStream_t stream, const char* name);

// out[c][r'] = in[perm.src(r')][c]  (perm.dk == 0: plain transpose)
int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t stream, HeadPerm perm = HeadPerm{0, 0});
// w2[r'] = w[perm.src(r')] (rows of `cols` floats), b2[r'] = b[perm.src(r')]
int launch_permute_rows(const float* w, const float* b, float* w2, float* b2, int rows, int cols, HeadPerm perm,
                        hipStream_t stream);

// gemm_bf16.hip: same NT contract with split-bf16 (npass 3) / bf16 (npass 1) MFMAs; wplanes is a
// scratch of gemm_nt_bf16_wplane_bytes(N, K) bytes that receives the bf16 planes of W
int launch_gemm_nt_bf16(int amode, int emode, int npass, const NTArgs& a, void* wplanes, hipStream_t stream,
                        const char* name);
size_t gemm_nt_bf16_wplane_bytes(int N, int K);
int launch_gemm_tn_bf16(int npass, const TNArgs& a, hipStream_t stream, const char* name);

// additive-attention forward (pool.hip: fp32 MFMA; gemm_bf16.hip: split-bf16 main loop)
struct AddFwdArgs {
    NTArgs g;              // A = ctx [M,d], W = w_add [q,d], bias = b_add, N = q, K = d
    const float* qv;       // [q]
    float* T;              // [M,q] or null
    float* wout;           // [M]   or null
    float* out;            // [n_seq, d]
    int S, d;
    const uint8_t* mask;   // optional [M]: masked_fill(mask == 0, -1e9) before the softmax (nrms_v1.py:100-101)
};
int launch_addattn_fwd_bf16(int npass, const AddFwdArgs& a, void* wplanes, hipStream_t stream);
// out[c] += sum_r partial[r][c] in a fixed order (pool.hip)
int launch_colsum_add(const float* partial, int rows, int cols, float* out, hipStream_t stream);

// fused16.hip: fp16 mode.  One wavefront per sequence; weight planes prepared once per call.
struct Fused16Layout { int KP, DP, QP; size_t wqkv16, wadd16, bqkv32, badd32, qv32, total; };   // byte offsets into `planes`
Fused16Layout fused16_layout(int d, int h, int q);
bool fused16_supported(int S, int d, int h, int q, const char** why);
int launch_prep16(int d, int h, int q, const float* w_qkv, const float* b_qkv, const float* w_add, const float* b_add,
                  const float* q_vec, void* planes, hipStream_t stream);
// x16[r, 0:KP] = fp16(table[ids[t]] * keep / (1-p)), t = live[r] for r < *n_live (live == null: t = r < M)
int launch_gather16(long M, int d, int KP, const int64_t* ids, const int* live, const int* n_live, const float* table,
                    const Dropout& drop, void* x16, hipStream_t stream);
int launch_cast16(long M, int d, int KP, const float* x, void* x16, hipStream_t stream);
struct Fused16Fwd {
    int n_seq, S, d, h, q;
    const void* planes;       // fused16_layout(d, h, q).total bytes, filled by launch_prep16
    const void* x16;          // [rows][KP] fp16
    const int* pos;           // token -> x16 row (-1: the padding token's row, index *n_rows) or null (row = token)
    const int* n_rows;        // device: number of compact rows (required with pos)
    const int64_t* ids;       // non-null: sequences whose ids are all 0 take the closed form (padding row is zero)
    const int* order;         // optional [3][n_seq] from launch_title_order(n_classes = 3) (+ order_cnt): long, all-padding, short titles
    const int* order_cnt;
    void* ctx16;              // [n_seq*S][DP] fp16 (always written: the kernel reads it back)
    void* t16;                // [n_seq*S][QP] fp16 or null (inference)
    float* w;                 // [n_seq*S] or null
    float* out;               // [n_seq][d]
    Dropout drop;             // context dropout
};
int launch_fused_fwd16(const Fused16Fwd& f, hipStream_t stream);
// fused16_v1.hip: the same for heads wider than 32 with the output projection (nrms_v1's news encoder); attn16 =
// [n_seq][20][32][16] fp16, the head concatenation in the operand order of the W_O tiles
bool fused16v1_supported(int S, int d, int h, int q, const char** why);
size_t fused16v1_planes_bytes(int d, int h, int q);
int launch_prep16v1(int d, int h, int q, const float* w_qkv, const float* b_qkv, const float* w_o, const float* b_o,
                    const float* w_add, const float* b_add, const float* q_vec, void* planes, hipStream_t stream);
int launch_fused_fwd16v1(const Fused16Fwd& f, int h, void* attn16, hipStream_t stream);
// order [n_classes][n_seq] ints, cnt title_order_cnt_ints(n_seq) ints (cnt[0 .. n_classes) = the list sizes).  n_classes 2: titles
// with a real token | all-padding titles; 3: long | all-padding | short-prefix titles (see fused16.hip)
size_t title_order_cnt_ints(int n_seq);
int launch_title_order(int n_seq, int S, const int64_t* ids, int* order, int* cnt, hipStream_t stream, int n_classes = 2);

// fused16_bwd.hip: the backward of the above.  `workspace` holds the backward weight planes, dout16, dZ16, dQKV16, the
// per-workgroup column sums and the split-M partial slabs (fused16_bwd_layout(M, n_seq).total bytes).
struct Fused16BwdLayout { int n_wg, tn_splits_qkv, tn_splits_add; size_t btiles, xtiles, qv16, bqkv32, dout16, dz16, dctx16, dqkv16, red, maps, scale, partial, total; };
Fused16BwdLayout fused16_bwd_layout(long M, int n_seq);
struct Fused16Bwd {
    int n_seq, S, d, h, q;
    void* workspace;
    const float* w_qkv; const float* b_qkv; const float* w_add; const float* q_vec;      // reference layout, fp32
    const void* x16;          // forward: [rows][KP]
    const int* pos;           // token -> row (x16 and dQKV16), -1 = padding token; null: row = token
    const int* n_rows_dev;    // number of rows (device) when pos != null, else null (= n_seq * S)
    const int64_t* ids;       // non-null: all-padding sequences take the closed form
    const int* order; const int* order_cnt;     // launch_title_order(n_classes = 3) lists, or null
    const void* ctx16; const void* t16; const float* w;         // forward activations
    const float* dout;        // [n_seq][d]
    float loss_scale;         // > 0: fixed power of two the fp16 gradients are carried multiplied by; <= 0: chosen on the
                              // device from max |dout| (loss_scale_from_max)
    Dropout drop;             // context dropout
    float* dw_qkv; float* db_qkv; float* dw_add; float* db_add; float* dq_vec;           // accumulated
    float* dx;                // [rows][d] fp32 (rows as pos / tokens): gradient w.r.t. the encoder input -- or, with dx_fp16,
                              // fp16 [rows][KP] STILL multiplied by the loss scale (the embedding scatter divides: *sc_out)
    bool dx_fp16;
    const float** sc_out;     // receives the device address of {scale, 1 / scale} (optional)
    bool defer_join;          // leave the two weight-gradient GEMMs running on their helper streams when the call returns
                              // (dx is complete in stream order); fused_bwd16_join() orders them before later work
    // nrms_v1's news encoder (launch_fused_bwd16v1): the output projection, the forward's head concatenation, their gradients
    const float* w_o; const void* attn16; float* dw_o; float* db_o;
};
int launch_fused_bwd16(const Fused16Bwd& f, hipStream_t stream);
// fused16_v1_bwd.hip: the backward of fused16_v1.hip (needs pos / ids / order: the padding-skipping path); workspace =
// fused16v1_bwd_bytes(M, n_seq, h) bytes
size_t fused16v1_bwd_bytes(long M, int n_seq, int h);
int launch_fused_bwd16v1(const Fused16Bwd& f, hipStream_t stream);
int fused_bwd16_join(hipStream_t stream);

// user64.hip: the split-bf16 user encoder (sequences of 33..64 rows) as one kernel per direction (NRMS_FLAG_FUSED_SEQ64)
bool user64_supported(int S, int d, int h, int q, const char** why);
size_t user64_qkv_bytes(int n_seq, int h);             // acts.qkv under the flag: operand fragments instead of [M, 3d] floats
size_t user64_fwd_planes_bytes(int h);                 // forward weight planes (acts.scratch)
size_t user64_bwd_planes_bytes(int h);                 // backward weight planes (workspace)
size_t user64_dq_partial_floats(int n_seq);
int launch_user64_fwd(int n_seq, int S, int d, int h, int q, const float* x, const float* w_qkv, const float* b_qkv,
                      const float* w_add, const float* b_add, const float* q_vec, void* planes, float* ctx, float* t, float* w,
                      void* qkv, float* out, bool train, hipStream_t stream);
// leaves ds [M] and dqkv [M, 3d] (head-major) for the weight-gradient / dX GEMMs; d(q_vec) is accumulated
int launch_user64_bwd(int n_seq, int S, int d, int h, int q, const float* w_add, const float* q_vec, void* planes, const float* dout,
                      const float* t, const float* w, const void* qkv, float* ds, float* dq_partial, float* dq_vec, float* dqkv,
                      hipStream_t stream);

// embed.hip
// dst[i] = src[i] if 0 <= src[i] < vocab else 0; *n_bad += ids replaced (dst may alias src)
int launch_sanitize_ids(long n, const void* src, bool src_is_int32, int64_t* dst, int vocab, int* n_bad, hipStream_t stream);
// keys[t] = 64-bit hash of the L word ids of title t
int launch_title_dedup(long n, int L, const int64_t* ids, int* table, long table_size, int* inverse, int* rep_rows,
                       int* n_unique, hipStream_t stream);
int launch_gather_dropout(long M, int d, const int64_t* ids, const float* table, const Dropout& drop, float* x,
                          hipStream_t stream);
// x[r, :] = table[ids[live[r]], :] * keep(live[r], :) / (1 - p) for the compact rows r < *n_live
int launch_gather_dropout_compact(long M, int d, const int64_t* ids, const int* live, const int* n_live,
                                  const float* table, const Dropout& drop, float* x, hipStream_t stream);
// out[m, :] = row[:] for every token m of the [n_seq, S] id matrix with ids[m] == 0 (n % 4 == 0); with
// skip_all_pad, sequences without any real token are left untouched
int launch_fill_pad_rows(long n_seq, int S, int n, const int64_t* ids, const float* row, float* out, bool skip_all_pad,
                         hipStream_t stream);
// live[0 .. *n_live) = the token positions m with ids[m] != 0 in ascending order; pos (optional) = inverse map
// (-1 for padding tokens); scratch: compact_scratch_ints(M) ints
size_t compact_scratch_ints(long M);
int launch_compact_live_rows(long M, const int64_t* ids, int* live, int* pos, int* n_live, int* scratch,
                             hipStream_t stream);
// grouped (atomic-free) scatter of a COMPACT dx; scratch: scatter_grouped_scratch_ints(M, V) ints
size_t scatter_grouped_scratch_ints(long M, int V);
// dx: fp32 [rows][d], or (dx_fp16) fp16 [rows][ldx] multiplied by the loss scale whose reciprocal is sc[1] (device)
int launch_scatter_grouped(long M, int V, int d, const int64_t* ids, const int* live, const int* n_live, const void* dx,
                           const Dropout& drop, float* dtable, int* scratch, hipStream_t stream, bool dx_fp16 = false, int ldx = 0,
                           const float* sc = nullptr, bool prepared = false);
// the id-only part of the above (histogram of the live tokens' ids, offsets, placement): `prepared` = it has run on `scratch`
int launch_scatter_prepare(long M, int V, const int64_t* ids, const int* live, const int* n_live, int* scratch, hipStream_t stream);
// dx has one row per token; the live tokens' rows are scatter-added (float atomics)
int launch_scatter_dense_rows(long M, int d, const int64_t* ids, const int* live, const int* n_live, const float* dx,
                              const Dropout& drop, float* dtable, hipStream_t stream);
// same as launch_scatter_dropout for a COMPACT dx: row r of dx belongs to token live[r], r < *n_live
int launch_scatter_dropout_compact(long M, int d, const int64_t* ids, const int* live, const int* n_live, const float* dx,
                                   const Dropout& drop, float* dtable, hipStream_t stream);

}  // namespace nrms
