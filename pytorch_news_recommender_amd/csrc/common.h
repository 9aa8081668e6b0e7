// Internal helpers shared by the gfx950 kernels of libnrms_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/nrms_hip.h"

namespace nrms {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------
// f32-input MFMA (exact fp32, k-ordered fma chain; 64 FLOP/clk/SIMD on gfx950).
// 16x16x4 : A lane l = A[i=l&15][k=l>>4], B lane l = B[k=l>>4][j=l&15],
//           D reg r = D[row=(l>>4)*4+r][col=l&15].
// 32x32x2 : A lane l = A[i=l&31][k=l>>5], B lane l = B[k=l>>5][j=l&31],
//           D reg r = D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// row of the 32x32 accumulator held in register r by lane-half hh
__device__ __forceinline__ int crow32(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// ---------------------------------------------------------------------------------------
// Counter-based dropout RNG: Philox4x32-7 keyed by the per-step seed; the counter is
// (group index lo, hi, site, 0) where a group is 4 consecutive elements of the flattened
// [rows, d] activation.  The backward regenerates the forward's mask from the same counter,
// so no mask is stored.  keep <=> u32 >= p * 2^32.
// ---------------------------------------------------------------------------------------
struct Keep4 { bool k[4]; };

__host__ __device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2,
                                                      uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__host__ __device__ __forceinline__ void philox4x32_7(uint64_t seed, uint64_t group, uint32_t site,
                                                      uint32_t out[4]) {
    uint32_t c0 = (uint32_t)group, c1 = (uint32_t)(group >> 32), c2 = site, c3 = 0x9E3779B9u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t <= 0.0) return 0u;
    if (t >= 4294967295.0) return 4294967295u;
    return (uint32_t)t;
}

// scale factors (0 or 1/(1-p)) for elements [4*group, 4*group+3] of a dropout site
__device__ __forceinline__ f32x4 dropout_scale4(uint64_t seed, uint32_t site, uint64_t group,
                                                uint32_t thresh, float inv_keep) {
    uint32_t r[4];
    philox4x32_7(seed, group, site, r);
    f32x4 s;
    s[0] = r[0] >= thresh ? inv_keep : 0.f;
    s[1] = r[1] >= thresh ? inv_keep : 0.f;
    s[2] = r[2] >= thresh ? inv_keep : 0.f;
    s[3] = r[3] >= thresh ? inv_keep : 0.f;
    return s;
}
// 16-bit fields: ONE Philox call decides 8 elements (field k of the call = bits 16 (k & 1) .. of word k >> 1; keep <=> field
// >= p * 2^16).  The 32-bit integer multiplies of Philox are quarter rate: in the fused fp16 kernels the context dropout
// was a tenth of the forward.  P(drop) is p rounded down to a multiple of 2^-16.
__host__ __device__ __forceinline__ uint32_t drop_threshold16(float p) {
    const double t = (double)p * 65536.0;
    return t <= 0.0 ? 0u : (t >= 65535.0 ? 65535u : (uint32_t)t);
}
__device__ __forceinline__ void dropout_scale8(uint64_t seed, uint32_t site, uint64_t group8, uint32_t thresh16, float inv_keep,
                                               float (&s)[8]) {
    uint32_t r[4];
    philox4x32_7(seed, group8, site, r);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        s[2 * w] = (r[w] & 0xFFFFu) >= thresh16 ? inv_keep : 0.f;
        s[2 * w + 1] = (r[w] >> 16) >= thresh16 ? inv_keep : 0.f;
    }
}
// single element (row-major [rows,d], d % 4 == 0)
__device__ __forceinline__ float dropout_scale1(uint64_t seed, uint32_t site, uint64_t elem,
                                                uint32_t thresh, float inv_keep) {
    uint32_t r[4];
    philox4x32_7(seed, elem >> 2, site, r);
    return r[elem & 3] >= thresh ? inv_keep : 0.f;
}

// Branch-free tanh for the split-bf16 modes: 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32 (absolute error
// ~1e-7, saturates correctly), and the odd Taylor polynomial where that form cancels (|x| < 0.12, truncation
// < 1e-10).  libm's tanhf is ~45 instructions with two branches per call; the additive-attention epilogue
// makes 104 calls per lane and tile.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.0f * x);
    const float r = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    const float x2 = x * x;
    const float p = x * (1.0f + x2 * (-0.33333333f + x2 * (0.13333333f + x2 * -0.053968254f)));
    return fabsf(x) < 0.12f ? p : r;
}

// Head-major column order of the Q/K/V projection.  The reference's W_Q, W_K, W_V stack gives output
// column n = which * d + head * d_k + j (which = 0,1,2 for Q,K,V; model/nrms_v0.py:53-58 splits the heads
// afterwards).  The HIP path stores the projection as n' = head * 3 d_k + which * d_k + j, so that one
// head's Q, K and V slices are a single contiguous 3 d_k block per row (attention.hip).  It is a pure
// relabelling of GEMM columns: the weight rows are permuted when they are staged, the weight gradient
// rows are permuted back when the partial sums are reduced.  dk == 0: identity.
struct HeadPerm {
    int dk, h;
    __host__ __device__ __forceinline__ int src(int np) const {       // n' -> n
        if (dk == 0) return np;
        const int head = np / (3 * dk), r = np - head * 3 * dk;
        const int which = r / dk, j = r - which * dk;
        return which * h * dk + head * dk + j;
    }
};

struct Dropout {
    uint64_t seed;
    uint32_t thresh;     // 0 => disabled
    float inv_keep;
    uint32_t thresh16;   // the same probability for the 16-bit-field scheme (dropout_scale8)
    // site 2 (attention probabilities) of a COMPACTED batch (nrms_encoder_desc::seq_index): sequence r of the call counts as
    // sequence seq_index[r] of the full batch, so the decisions are those of the uncompacted call
    const int* seq_index;
    int seq_h;           // heads per sequence (unit = seq * h + head)
    __device__ __forceinline__ long unit_of(long unit) const {
        if (seq_index == nullptr) return unit;
        const long seq = unit / seq_h;
        return (long)seq_index[seq] * seq_h + (unit - seq * seq_h);
    }
};

inline Dropout make_dropout(uint64_t seed, float p) {
    Dropout d;
    d.seed = seed;
    d.thresh = drop_threshold(p);
    d.thresh16 = drop_threshold16(p);
    d.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    d.seq_index = nullptr;
    d.seq_h = 1;
    return d;
}

// wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------
// host side: error text + optional per-kernel event timing
// ---------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- helper streams of the backward (capi.hip).  One SET per (device, caller stream), created the first time that stream
// runs a backward: the weight-gradient GEMMs fork from the caller's stream onto the set's streams and are joined back into it
// inside the same C-ABI call (or by the deferred join).  Two engines on two streams -- or on two devices -- of one process
// get two independent sets; nothing is shared between caller streams.  nullptr = keep everything on the caller's stream
// (NRMS_NO_SIDE_STREAMS, creation failure, or more than 32 distinct caller streams).
struct SideSet {
    hipStream_t s[2] = {nullptr, nullptr};
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool pending = false;          // a deferred join is outstanding (NRMS_FLAG_DEFER_WQKV, fp16 mode)
};
SideSet* side_streams_for(hipStream_t caller);
// `to` continues from what is enqueued on `from` so far (event e of the set); NRMS_ELAUNCH when the runtime refuses
int side_order(SideSet* ss, int e, hipStream_t from, hipStream_t to, const char* what);
// On scope exit (an error return in the middle of a backward included) the caller's stream waits for whatever the set's
// streams have been given, so the caller may reuse the workspace they read; disarm() when the normal path has joined them
// itself or leaves them running on purpose (deferred join).
struct SideJoinGuard {
    SideSet* ss; hipStream_t caller; bool armed;
    SideJoinGuard(SideSet* set, hipStream_t c) : ss(set), caller(c), armed(set != nullptr) {}
    void disarm() { armed = false; }
    ~SideJoinGuard();
};

struct TimingScope {
    TimingScope(const char* name, hipStream_t s);
    ~TimingScope();
    bool active;
    int slot;
    hipStream_t stream;
};

#define NRMS_REQUIRE(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            ::nrms::set_error(__VA_ARGS__);             \
            return NRMS_EINVAL;                         \
        }                                               \
    } while (0)

__host__ __device__ inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace nrms
