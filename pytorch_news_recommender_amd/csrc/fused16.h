// Shared device helpers of the fp16 mode (fused16.hip: forward; fused16_bwd.hip: backward + its GEMMs).
#pragma once
#include "gemm.h"

namespace nrms {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma32h(const h8& a, const h8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// registers 8s..8s+7 of a 32x32 accumulator as the fp16 operand of k-step s (rows of the accumulator = k)
__device__ __forceinline__ h8 acc_frag(const f32x16& x, int s) {
    h8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (_Float16)x[8 * s + j];
    return r;
}

// padded feature index held by register r of lane-half hh (accumulator row), and the P16 memory position of it
__device__ __forceinline__ int p16_pos(int fpad) {           // natural padded feature -> position inside its row
    const int b = fpad >> 4, t = fpad & 15;
    const int hh = (t >> 2) & 1, j = ((t >> 3) << 2) | (t & 3);
    return 16 * b + 8 * hh + j;
}

// Timing experiments ("what does the kernel cost without this store?") are compiled in only with -DNRMS_F16_EXPERIMENTS
// and then switched by the NRMS_F16_DBG environment variable; they make results WRONG, so a normal build has none of them.
#ifdef NRMS_F16_EXPERIMENTS
#define F16_DBG(flags, bit) ((flags) & (bit))
#else
#define F16_DBG(flags, bit) 0
#endif

constexpr int F16_WAVES = 4;            // waves (= sequences) per workgroup; two workgroups share a CU (one wave per SIMD each),
constexpr int F16_THREADS = 64 * F16_WAVES;   // so the two waves of a SIMD are in different phases: one's MFMAs cover the other's VALU / waits
// Pitches are compile-time constants (every inner loop fully unrolled, no guards: a guarded MFMA costs a branch
// and a full LDS wait each).  Smaller models run zero padded at these sizes.
constexpr int F16_KS = 20;            // k-steps of 16 input features held in registers: KP = 320 >= d
constexpr int F16_CS = 20;            // ctx k-steps: DP = 320 >= 32 h
constexpr int F16_QT = 7;             // q tiles of 32: QP = 224 >= q
constexpr int F16_KP = 16 * F16_KS, F16_DP = 16 * F16_CS, F16_QP = 32 * F16_QT;
constexpr int F16_PITCH = (F16_KP + 8) * 2;              // LDS row pitch in bytes: +16 B => conflict-free b128 reads
constexpr int F16_SLOT = 32 * F16_PITCH;
constexpr int F16_RC = F16_KP / 8;                       // 16-byte chunks per tile row
static_assert(F16_KP == F16_DP, "one tile geometry for the head tiles and the additive tiles");
constexpr int F16_STG = (32 * F16_RC + F16_THREADS - 1) / F16_THREADS;      // staging chunks (16 B) per thread and tile
constexpr bool F16_STG_EXACT = 32 * F16_RC == F16_STG * F16_THREADS;

// ---- "fragment order" of the per-token activations the fused kernels exchange through HBM (ctx16, d(ctx)16, T16, dZ16):
//      [sequence][32-row block][k-step of 16 columns][token 0..31][16 columns]   (fp16)
// A lane (token l32, half hh) touches 16 B per k-step, and the 64 lanes of a wave-instruction touch ONE contiguous KiB
// (token-major rows gave 32 separate 32-byte pieces per instruction).  Rows of tokens beyond the sequence exist and
// hold zeros, so the weight-gradient GEMM can stage a block as 32 contiguous rows.
__device__ __forceinline__ long frag_off(long seq_block, int c16, int s, int l32, int hh) {
    return (seq_block * c16 + s) * 512 + l32 * 16 + 8 * hh;
}

// ---- the weight-tile ring shared by the forward and the backward kernel: tile n lives in LDS slot n % 3; while
// tile n is consumed, tile n + 2 travels global -> registers -> LDS.  One barrier per tile.
struct TileRing {
    char* smem;
    const _Float16* src;          // tiles are contiguous [n_tiles][32][F16_KP]
    int n_tiles, tid, l32, hh;
    int dbg;
    h8 stg[F16_STG];
    __device__ __forceinline__ void load(int n) {
        if (n >= n_tiles) return;
        if (F16_DBG(dbg, 1)) return;
        // uniform base (scalar registers) + ONE 32-bit lane offset for all chunks: per-chunk 64-bit lane addresses were ten
        // registers the backward kernels spilled and reloaded in front of every tile
        const char* t = reinterpret_cast<const char*>(src) + (long)n * (32 * F16_KP * 2);
        const unsigned off = (unsigned)tid * 16u;
#pragma unroll
        for (int i = 0; i < F16_STG; ++i)
            if (F16_STG_EXACT || i + 1 < F16_STG || tid + F16_THREADS * i < 32 * F16_RC)
                stg[i] = *reinterpret_cast<const h8*>(t + (long)F16_THREADS * 16 * i + off);
    }
    __device__ __forceinline__ void store(int n) {
        if (n >= n_tiles) return;
        if (F16_DBG(dbg, 1)) return;
        char* dst = smem + (n % 3) * F16_SLOT;
#pragma unroll
        for (int i = 0; i < F16_STG; ++i) {
            const int c = tid + F16_THREADS * i;
            if (F16_STG_EXACT || i + 1 < F16_STG || c < 32 * F16_RC) {
                const int row = c / F16_RC, col = c - row * F16_RC;
                *reinterpret_cast<h8*>(dst + row * F16_PITCH + col * 16) = stg[i];
            }
        }
    }
    // row l32 of tile n, k-step s: 8 consecutive k for this lane half
    __device__ __forceinline__ h8 frag(int n, int s) const {
        return *reinterpret_cast<const h8*>(smem + (n % 3) * F16_SLOT + l32 * F16_PITCH + (16 * s + 8 * hh) * 2);
    }
};

// ---- the same ring filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass, and a tile has
// two whole steps to arrive instead of one).  A DMA instruction writes 64 lanes x 16 B = one contiguous KiB of LDS, so the
// tile is stored in HBM in the order its fragments are read: [k-step s][lane half hh][row l32][8 halves] -- the fragment
// read of a wave is then one contiguous KiB too (conflict-free without padding).  Tile n + 2 is issued at the top of step
// n; at the end of step n every wave waits until only those five newest loads are outstanding (in-order counter: tile
// n + 1 has landed) and meets the others at a raw s_barrier (__syncthreads() would drain the DMA with vmcnt(0)).
constexpr int F16_SLOT_DMA = 32 * F16_KP * 2;
static_assert(F16_SLOT_DMA == 5 * F16_WAVES * 1024, "five 1-KiB DMA pieces per wave and tile");
// SLOTS = 3: tile n + 2 is issued during step n (two steps to land).  SLOTS = 2: tile n + 1 is issued in the first MFMA
// groups of step n and awaited at its end -- 40 KB of LDS per workgroup instead of 60, which is what lets a third
// workgroup share the CU.
template <int SLOTS>
struct TileRingDMA {
    static constexpr int AHEAD = SLOTS - 1;
    char* smem;
    const _Float16* src;          // tiles are contiguous [n_tiles][KS][2][32][8]
    int n_tiles, wave, lane, l32, hh;
    __device__ __forceinline__ void load(int n) {
        if (n >= n_tiles) return;
        const char* g = reinterpret_cast<const char*>(src) + (long)n * F16_SLOT_DMA + wave * 1024 + lane * 16;
        char* l = smem + (n % SLOTS) * F16_SLOT_DMA + wave * 1024;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * (F16_WAVES * 1024)),
                                             (__attribute__((address_space(3))) void*)(l + i * (F16_WAVES * 1024)), 16, 0, 0);
    }
    // piece i (0..4) of tile n alone: issued between MFMA groups, where the instruction's issue cost hides under the
    // matrix pipe (five back-to-back DMA issues in front of a tile's MFMAs cost as much as the staging they replaced)
    __device__ __forceinline__ void load_piece(int n, int i) {
        if (n >= n_tiles) return;
        const char* g = reinterpret_cast<const char*>(src) + (long)n * F16_SLOT_DMA + wave * 1024 + lane * 16 + i * (F16_WAVES * 1024);
        char* l = smem + (n % SLOTS) * F16_SLOT_DMA + wave * 1024 + i * (F16_WAVES * 1024);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
    // the share of tile `n`'s five pieces that goes behind MFMA group g
    __device__ __forceinline__ void load_group(int n, int g) {
        if (SLOTS == 3) { load_piece(n, g); return; }
        if (g == 0) { load_piece(n, 0); load_piece(n, 1); }
        if (g == 1) { load_piece(n, 2); load_piece(n, 3); }
        if (g == 2) load_piece(n, 4);
    }
    // The same with the LDS slot chosen by the STEP and the source by the TILE: a kernel whose tile sequence revisits tiles
    // (fused_fwd16p: the additive tiles once per title of the wave) walks steps 0, 1, 2, ... and names the tile of each.
    __device__ __forceinline__ void load_at(int step, int tile) {
        const char* g = reinterpret_cast<const char*>(src) + (long)tile * F16_SLOT_DMA + wave * 1024 + lane * 16;
        char* l = smem + (step % SLOTS) * F16_SLOT_DMA + wave * 1024;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * (F16_WAVES * 1024)),
                                             (__attribute__((address_space(3))) void*)(l + i * (F16_WAVES * 1024)), 16, 0, 0);
    }
    __device__ __forceinline__ void load_piece_at(int step, int tile, int i) {
        const char* g = reinterpret_cast<const char*>(src) + (long)tile * F16_SLOT_DMA + wave * 1024 + lane * 16 + i * (F16_WAVES * 1024);
        char* l = smem + (step % SLOTS) * F16_SLOT_DMA + wave * 1024 + i * (F16_WAVES * 1024);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
    __device__ __forceinline__ void store(int) {}
    // end of the step that consumed tile n: tile n + 1 is complete in LDS for every wave after this.  `younger` = vector
    // memory instructions this wave issued in this step AFTER the five pieces of tile n + 2 (stores of the step's results):
    // the counter is in order, so tile n + 1 is complete once at most 5 + younger operations are outstanding.  It must never
    // be an over-estimate (0 is always safe).
    template <int YOUNGER = 0>
    __device__ __forceinline__ void step_barrier(int n) {
        static_assert(YOUNGER == 0 || YOUNGER == 2 || YOUNGER == 4 || YOUNGER == 8, "add the immediate below");
        if (SLOTS == 3 && n + 2 < n_tiles) {
            if (YOUNGER == 0) __asm__ volatile("s_waitcnt vmcnt(5)" ::: "memory");
            if (YOUNGER == 2) __asm__ volatile("s_waitcnt vmcnt(7)" ::: "memory");
            if (YOUNGER == 4) __asm__ volatile("s_waitcnt vmcnt(9)" ::: "memory");
            if (YOUNGER == 8) __asm__ volatile("s_waitcnt vmcnt(13)" ::: "memory");
        } else {
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __device__ __forceinline__ h8 frag(int n, int s) const {
        return *reinterpret_cast<const h8*>(smem + (n % SLOTS) * F16_SLOT_DMA + ((2 * s + hh) * 32 + l32) * 16);
    }
};
// position (in halves) of element (row r, column k) of a tile in that order
__host__ __device__ __forceinline__ int dma_tile_pos(int r, int k) { return (((k >> 4) * 2 + ((k >> 3) & 1)) * 32 + r) * 8 + (k & 7); }

// acc += (W tile n) x (register operand), 20 k-steps as 5 groups of 4: the weight fragments of group g + 1 are read
// from LDS while the MFMAs of group g issue (two named register sets; the scheduling barriers keep hipcc from hoisting
// all twenty reads to the top, which spills).  W_IS_A: acc = W x^T (features x tokens); else acc = x W^T.
struct NoPrefetch { __device__ __forceinline__ void operator()(int) const {} };
// `between(g)` runs after the MFMAs of group g have been issued (the place for one DMA piece of a later tile)
template <bool W_IS_A, class Ring, int NS, class Between = NoPrefetch>
__device__ __forceinline__ void tile_mma(f32x16& acc, const Ring& ring, int n, const h8 (&op)[NS], Between between = Between()) {
    static_assert(NS % 4 == 0, "k-steps in groups of 4");
    h8 wa[4], wb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wa[i] = ring.frag(n, i);
#pragma unroll
    for (int g = 0; g < NS / 4; ++g) {
        if (g + 1 < NS / 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (g & 1) wa[i] = ring.frag(n, 4 * (g + 1) + i);
                else wb[i] = ring.frag(n, 4 * (g + 1) + i);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const h8& w = (g & 1) ? wb[i] : wa[i];
            acc = W_IS_A ? mfma32h(w, op[4 * g + i], acc) : mfma32h(op[4 * g + i], w, acc);
        }
        between(g);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// r[4g + e] = v[8g + 4hh + e]: a per-ROW vector (bias) in the accumulator's register order.  Loaded at the top of a
// tile step and ADDED after the MFMAs: as the accumulator's initial value it would put a global-load latency in
// front of every tile.
__device__ __forceinline__ f32x16 rows_of(const float* v, int hh) {
    f32x16 r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(v + 8 * g + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[4 * g + e] = bb[e];
    }
    return r;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 0.f;
    return r;
}


}  // namespace nrms
