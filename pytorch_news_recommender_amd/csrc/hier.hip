// Integer / index side of a HieRec-style hierarchical interest model (BASELINE configs[3]; no reference
// implementation: PARITY UNPINNED -- the checker is oracle/segpool_oracle.py::hierarchical_interest / hierarchical_scores):
//   nrms_hier_tree_build   a user's clicked news grouped by sub-topic, the sub-topic groups by topic, the topic groups per user,
//                          as the three index lists nrms_segment_pool_fwd / _bwd aggregate over (csrc/segpool.hip)
//   nrms_hier_add_embedding_fwd / _bwd   interest = aggregate + the (sub-)topic's embedding; table gradient without atomics
//   nrms_hier_match, nrms_hier_score_fwd / _bwd   hierarchical matching: a candidate against the user's interest in ITS sub-topic
//                          and topic (weighted by the share of the user's clicks there) and against the overall interest
// Everything here is HBM-bound byte / index work: one wavefront per user (a history has at most 64 slots = one lane each,
// grouping by wave ballots, no sorting), fixed per-user strides (slot b * H + g), a single-block scan for the packed lists.
#include "gemm.h"

namespace nrms {

constexpr int HIER_WPB = 4;

// ---- per user (one wave): sub-topic groups in order of first occurrence, topic groups of those in order of first occurrence
struct HierTreeArgs {
    int B, H;
    const uint8_t* valid;       // [B, H]
    const int64_t* topic;       // [B, H]
    const int64_t* sub;         // [B, H]
    int* l1_sub; int* l1_top; int* l1_cnt;      // [B * H] per sub-topic group slot (b * H + g): ids, clicks (0 = empty slot)
    int* l2_top; int* l2_cnt;                    // [B * H] per topic group slot (b * H + tg)
    int* n_valid;                                // [B]
    unsigned long long* m1;     // [B * H] member mask (over the H slots) of sub-topic group g
    unsigned long long* m2;     // [B * H] member mask (over the sub-topic group slots) of topic group tg
    int* counts;                // [3][B]: members per user at the three levels (valid slots, sub-topic groups, topic groups)
};

__global__ __launch_bounds__(64 * HIER_WPB) void hier_tree_kernel(HierTreeArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * HIER_WPB + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const int H = a.H;
    const bool v = lane < H && a.valid[(long)b * H + lane] != 0;
    const int st = v ? (int)a.sub[(long)b * H + lane] : -1;
    const int tp = v ? (int)a.topic[(long)b * H + lane] : -1;
    unsigned long long rem = __ballot(v);
    const int nv = __popcll(rem);
    // level 1: groups by sub-topic id, in order of first occurrence
    int g = 0;
    int my_g = -1;                                   // (lane k: its group)
    int g_sub = -1, g_top = -1, g_cnt = 0;           // lane g holds group g's ids once assigned
    unsigned long long g_mask = 0ull;
    while (rem != 0ull) {
        const int lead = __ffsll((long long)rem) - 1;
        const int sv = __shfl(st, lead, 64), tv = __shfl(tp, lead, 64);
        const unsigned long long mem = __ballot(v && st == sv) & rem;
        if ((mem >> lane) & 1ull) my_g = g;
        if (lane == g) { g_sub = sv; g_top = tv; g_cnt = __popcll(mem); g_mask = mem; }
        rem &= ~mem;
        ++g;
    }
    const int n1 = g;
    if (lane < H) {
        const long s = (long)b * H + lane;
        a.l1_sub[s] = lane < n1 ? g_sub : -1;
        a.l1_top[s] = lane < n1 ? g_top : -1;
        a.l1_cnt[s] = lane < n1 ? g_cnt : 0;
        a.m1[s] = lane < n1 ? g_mask : 0ull;
    }
    // level 2: the sub-topic groups (lane = group slot) by topic id
    const bool v2 = lane < n1;
    unsigned long long rem2 = __ballot(v2);
    int tg = 0, t_top = -1, t_cnt = 0;
    unsigned long long t_mask = 0ull;
    while (rem2 != 0ull) {
        const int lead = __ffsll((long long)rem2) - 1;
        const int tv = __shfl(g_top, lead, 64);
        const unsigned long long mem = __ballot(v2 && g_top == tv) & rem2;
        int clicks = ((mem >> lane) & 1ull) ? g_cnt : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) clicks += __shfl_xor(clicks, o, 64);
        if (lane == tg) { t_top = tv; t_cnt = clicks; t_mask = mem; }
        rem2 &= ~mem;
        ++tg;
    }
    const int n2 = tg;
    if (lane < H) {
        const long s = (long)b * H + lane;
        a.l2_top[s] = lane < n2 ? t_top : -1;
        a.l2_cnt[s] = lane < n2 ? t_cnt : 0;
        a.m2[s] = lane < n2 ? t_mask : 0ull;
    }
    if (lane == 0) {
        a.n_valid[b] = nv;
        a.counts[b] = nv;
        a.counts[a.B + b] = n1;
        a.counts[2 * a.B + b] = n2;
    }
    (void)my_g;
}

// exclusive scans of the three per-user member counts (one block; B <= 2^20)
__global__ __launch_bounds__(1024) void hier_scan_kernel(int B, const int* counts, int* base) {
    __shared__ int part[1024];
    for (int level = 0; level < 3; ++level) {
        const int* c = counts + (long)level * B;
        int* o = base + (long)level * (B + 1);
        const int per = (B + 1023) / 1024;
        const int lo = threadIdx.x * per, hi = min(lo + per, B);
        int s = 0;
        for (int i = lo; i < hi; ++i) s += c[i];
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int i = 0; i < 1024; ++i) { const int t = part[i]; part[i] = run; run += t; }
            o[B] = run;
        }
        __syncthreads();
        int run = part[threadIdx.x];
        for (int i = lo; i < hi; ++i) { o[i] = run; run += c[i]; }
        __syncthreads();
    }
}

// the packed lists: level 1 rows b * H + k (history slots), level 2 rows b * H + g (sub-topic group slots), level 3 rows
// b * H + tg (topic group slots); segments: level 1 / 2 one per group SLOT b * H + g (empty slots: empty segments), level 3 one per user
struct HierEmitArgs {
    int B, H;
    const int* l1_cnt; const int* l2_cnt; const int* counts; const int* base;
    const unsigned long long* m1; const unsigned long long* m2;
    int* l1_ptr; int* l1_idx; int* l2_ptr; int* l2_idx; int* l3_ptr; int* l3_idx;
};

__global__ __launch_bounds__(64 * HIER_WPB) void hier_emit_kernel(HierEmitArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * HIER_WPB + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const int H = a.H, B = a.B;
    const int n1 = a.counts[B + b], n2 = a.counts[2 * B + b];
    const int base1 = a.base[b], base2 = a.base[(B + 1) + b], base3 = a.base[2 * (B + 1) + b];
    const long s = (long)b * H + lane;
    // level 1: segment slot g = lane; members in ascending slot order
    {
        const int cnt = lane < H ? __popcll(a.m1[s]) : 0;
        int off = cnt;                                // inclusive scan over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(off, o, 64); if (lane >= o) off += t; }
        const int start = base1 + off - cnt;
        if (lane < H) {
            a.l1_ptr[s] = start;
            unsigned long long m = a.m1[s];
            int w = start;
            while (m != 0ull) { const int k = __ffsll((long long)m) - 1; a.l1_idx[w++] = b * H + k; m &= m - 1ull; }
        }
        if (b == B - 1 && lane == 0) a.l1_ptr[(long)B * H] = a.base[B];
    }
    // level 2
    {
        const int cnt = lane < H ? __popcll(a.m2[s]) : 0;
        int off = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(off, o, 64); if (lane >= o) off += t; }
        const int start = base2 + off - cnt;
        if (lane < H) {
            a.l2_ptr[s] = start;
            unsigned long long m = a.m2[s];
            int w = start;
            while (m != 0ull) { const int k = __ffsll((long long)m) - 1; a.l2_idx[w++] = b * H + k; m &= m - 1ull; }
        }
        if (b == B - 1 && lane == 0) a.l2_ptr[(long)B * H] = a.base[(B + 1) + B];
    }
    // level 3: one segment per user over its topic group slots
    if (lane == 0) {
        a.l3_ptr[b] = base3;
        if (b == B - 1) a.l3_ptr[B] = a.base[2 * (B + 1) + B];
    }
    if (lane < n2) a.l3_idx[base3 + lane] = b * H + lane;
    (void)n1;
}

// u[slot][:] += table[id[slot]][:] for the occupied slots
__global__ __launch_bounds__(256) void hier_addemb_kernel(long n_slots, int d4, int n_ids, const int* id, const int* cnt, const float* table, float* u) {
    const long total = n_slots * d4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long s = i / d4;
        const int c = (int)(i - s * d4);
        const int r = id[s];
        if (cnt[s] <= 0 || r < 0 || r >= n_ids) continue;          // (an id outside the table: no embedding, as in the backward)
        f32x4 v = *reinterpret_cast<const f32x4*>(u + i * 4);
        v += *reinterpret_cast<const f32x4*>(table + ((long)r * d4 + c) * 4);
        *reinterpret_cast<f32x4*>(u + i * 4) = v;
    }
}

// dtable[r][:] += sum over the occupied slots with id == r of du[slot][:], in a fixed order (no atomics), two kernels:
//   partial[c][r][:] = the sum over the slots of chunk c, ascending (one workgroup per (row, chunk); a wave scans 64 slots by ballot)
//   dtable[r][:]    += sum over c ascending of partial[c][r][:]
__global__ __launch_bounds__(256) void hier_embgrad_kernel(long n_slots, long chunk, int d, const int* id, const int* cnt, const float* du, float* partial) {
    const int r = blockIdx.x;
    const long c0 = (long)blockIdx.y * chunk;
    const long c1 = c0 + chunk < n_slots ? c0 + chunk : n_slots;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};            // columns threadIdx.x + 256 j  (d <= 1024)
    for (long s0 = c0; s0 < c1; s0 += 64) {
        const long s = s0 + (threadIdx.x & 63);
        const bool hit = s < c1 && cnt[s] > 0 && id[s] == r;
        unsigned long long m = __ballot(hit);        // (the same in every wave: all waves read the same 64 slots)
        while (m != 0ull) {
            const int k = __ffsll((long long)m) - 1;
            const float* row = du + (s0 + k) * d;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int c = threadIdx.x + 256 * j; if (c < d) acc[j] += row[c]; }
            m &= m - 1ull;
        }
    }
    float* out = partial + ((long)blockIdx.y * gridDim.x + r) * d;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = threadIdx.x + 256 * j; if (c < d) out[c] = acc[j]; }
}

__global__ __launch_bounds__(256) void hier_embgrad_reduce_kernel(long n_elems, int n_chunks, const float* partial, float* dtable) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_elems) return;
    float acc = 0.f;
    for (int c = 0; c < n_chunks; ++c) acc += partial[(long)c * n_elems + i];
    dtable[i] += acc;
}

static int embgrad_chunks(long n_slots, int n_ids) {
    long c = 8192 / (n_ids > 0 ? n_ids : 1);
    c = c < 8 ? 8 : (c > 128 ? 128 : c);
    const long most = (n_slots + 63) / 64;           // at least 64 slots per chunk
    if (c > most) c = most > 0 ? most : 1;
    return (int)c;
}

// candidate (b, c) -> the user's sub-topic / topic group slot it falls into (-1: none) and the share of the user's clicks there
__global__ __launch_bounds__(256) void hier_match_kernel(int B, int C, int H, const int64_t* cand_topic, const int64_t* cand_sub, const int* l1_sub,
                                                         const int* l1_cnt, const int* l2_top, const int* l2_cnt, const int* n_valid,
                                                         int* sub_slot, float* sub_frac, int* top_slot, float* top_frac) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * C) return;
    const int b = (int)(i / C);
    const int cs = (int)cand_sub[i], ct = (int)cand_topic[i];
    const float inv = n_valid[b] > 0 ? 1.0f / (float)n_valid[b] : 0.f;
    int ss = -1, ts = -1;
    float sf = 0.f, tf = 0.f;
    for (int g = 0; g < H; ++g) {
        const long s = (long)b * H + g;
        if (ss < 0 && l1_cnt[s] > 0 && l1_sub[s] == cs) { ss = (int)s; sf = l1_cnt[s] * inv; }
        if (ts < 0 && l2_cnt[s] > 0 && l2_top[s] == ct) { ts = (int)s; tf = l2_cnt[s] * inv; }
    }
    sub_slot[i] = ss; sub_frac[i] = sf; top_slot[i] = ts; top_frac[i] = tf;
}

struct HierScoreArgs {
    int B, C, d;
    const float* cand;      // [B, C, d]
    const float* u1;        // [B * H, d] sub-topic interests
    const float* u2;        // [B * H, d] topic interests
    const float* ug;        // [B, d]
    const int* sub_slot; const float* sub_frac; const int* top_slot; const float* top_frac;
    const uint8_t* mask;    // [B, C] or null
    float ls, lt, lg;
    float* scores;          // forward
    const float* dscores;   // backward
    float* dcand; float* du1; float* du2; float* dug;     // backward: dcand overwritten; du1 / du2 accumulated; dug overwritten
};

// one wave per (user, candidate)
__global__ __launch_bounds__(64 * HIER_WPB) void hier_score_fwd_kernel(HierScoreArgs a) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * HIER_WPB + (threadIdx.x >> 6);
    if (i >= (long)a.B * a.C) return;
    const int b = (int)(i / a.C);
    const float* n = a.cand + i * a.d;
    const int ss = a.sub_slot[i], ts = a.top_slot[i];
    const float cs = ss >= 0 ? a.ls * a.sub_frac[i] : 0.f, ct = ts >= 0 ? a.lt * a.top_frac[i] : 0.f;
    const float* p1 = a.u1 + (long)(ss >= 0 ? ss : 0) * a.d;
    const float* p2 = a.u2 + (long)(ts >= 0 ? ts : 0) * a.d;
    const float* pg = a.ug + (long)b * a.d;
    float acc = 0.f;
    for (int k = lane; k < a.d; k += 64) acc += n[k] * (cs * p1[k] + ct * p2[k] + a.lg * pg[k]);
    acc = wave_sum(acc);
    if (lane == 0) a.scores[i] = (a.mask != nullptr && a.mask[i] == 0) ? -1e9f : acc;
}

// one wave per USER: its candidates in ascending order (several may fall into the same interest slot: no atomics)
__global__ __launch_bounds__(64 * HIER_WPB) void hier_score_bwd_kernel(HierScoreArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * HIER_WPB + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const float* pg = a.ug + (long)b * a.d;
    float* dg = a.dug + (long)b * a.d;
    for (int k = lane; k < a.d; k += 64) dg[k] = 0.f;
    for (int c = 0; c < a.C; ++c) {
        const long i = (long)b * a.C + c;
        float* dn = a.dcand + i * a.d;
        const bool live = a.mask == nullptr || a.mask[i] != 0;
        const float ds = live ? a.dscores[i] : 0.f;          // a masked slot passes no gradient
        const int ss = a.sub_slot[i], ts = a.top_slot[i];
        const float cs = ss >= 0 ? a.ls * a.sub_frac[i] : 0.f, ct = ts >= 0 ? a.lt * a.top_frac[i] : 0.f;
        const float* n = a.cand + i * a.d;
        const float* p1 = a.u1 + (long)(ss >= 0 ? ss : 0) * a.d;
        const float* p2 = a.u2 + (long)(ts >= 0 ? ts : 0) * a.d;
        float* d1 = a.du1 + (long)(ss >= 0 ? ss : 0) * a.d;
        float* d2 = a.du2 + (long)(ts >= 0 ? ts : 0) * a.d;
        for (int k = lane; k < a.d; k += 64) {
            dn[k] = ds * (cs * p1[k] + ct * p2[k] + a.lg * pg[k]);
            const float gn = ds * n[k];
            if (ss >= 0) d1[k] += cs * gn;
            if (ts >= 0) d2[k] += ct * gn;
            dg[k] += a.lg * gn;
        }
    }
}

}  // namespace nrms

using namespace nrms;

extern "C" size_t nrms_hier_tree_scratch_bytes(int32_t B, int32_t H) {
    if (B <= 0 || H <= 0 || H > 64) return 0;
    // m1, m2 (uint64 [B * H] each), counts [3][B], base [3][B + 1]
    return (size_t)2 * B * H * 8 + (size_t)(3 * B + 3 * (B + 1)) * 4 + 256;
}

extern "C" int nrms_hier_tree_build(int32_t B, int32_t H, const uint8_t* valid, const int64_t* topic, const int64_t* subtopic,
                                    int32_t* l1_ptr, int32_t* l1_idx, int32_t* l1_sub, int32_t* l1_top, int32_t* l1_cnt,
                                    int32_t* l2_ptr, int32_t* l2_idx, int32_t* l2_top, int32_t* l2_cnt, int32_t* l3_ptr,
                                    int32_t* l3_idx, int32_t* n_valid, void* scratch, size_t scratch_bytes, void* stream) {
    NRMS_REQUIRE(B >= 0 && H >= 1 && H <= 64 && (long)B * H < (1L << 30) && B <= (1 << 20), "hier_tree_build: B=%d H=%d (H <= 64)", B, H);
    if (B == 0) return NRMS_OK;
    NRMS_REQUIRE(valid && topic && subtopic && l1_ptr && l1_idx && l1_sub && l1_top && l1_cnt && l2_ptr && l2_idx && l2_top && l2_cnt &&
                 l3_ptr && l3_idx && n_valid && scratch, "hier_tree_build: null argument");
    NRMS_REQUIRE(scratch_bytes >= nrms_hier_tree_scratch_bytes(B, H) && ((uintptr_t)scratch & 7) == 0, "hier_tree_build: scratch too small / unaligned");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* m1 = (unsigned long long*)scratch;
    unsigned long long* m2 = m1 + (size_t)B * H;
    int* counts = (int*)(m2 + (size_t)B * H);
    int* base = counts + 3 * B;
    HierTreeArgs a{B, H, valid, topic, subtopic, l1_sub, l1_top, l1_cnt, l2_top, l2_cnt, n_valid, m1, m2, counts};
    TimingScope ts("hier_tree", s);
    hipLaunchKernelGGL(hier_tree_kernel, dim3(cdiv(B, HIER_WPB)), dim3(64 * HIER_WPB), 0, s, a);
    hipLaunchKernelGGL(hier_scan_kernel, dim3(1), dim3(1024), 0, s, B, counts, base);
    HierEmitArgs e{B, H, l1_cnt, l2_cnt, counts, base, m1, m2, l1_ptr, l1_idx, l2_ptr, l2_idx, l3_ptr, l3_idx};
    hipLaunchKernelGGL(hier_emit_kernel, dim3(cdiv(B, HIER_WPB)), dim3(64 * HIER_WPB), 0, s, e);
    return check_launch("hier_tree_build");
}

extern "C" int nrms_hier_add_embedding_fwd(int64_t n_slots, int32_t d, int32_t n_ids, const int32_t* id, const int32_t* cnt, const float* table,
                                           float* u, void* stream) {
    NRMS_REQUIRE(n_slots >= 0 && d > 0 && (d & 3) == 0 && n_ids > 0, "hier_add_embedding_fwd: n_slots=%ld d=%d n_ids=%d", (long)n_slots, d, n_ids);
    if (n_slots == 0) return NRMS_OK;
    NRMS_REQUIRE(id && cnt && table && u, "hier_add_embedding_fwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    long blocks = (n_slots * (d / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    TimingScope ts("hier_addemb", s);
    hipLaunchKernelGGL(hier_addemb_kernel, dim3((int)blocks), dim3(256), 0, s, (long)n_slots, d / 4, n_ids, id, cnt, table, u);
    return check_launch("hier_add_embedding_fwd");
}

extern "C" size_t nrms_hier_add_embedding_bwd_workspace_bytes(int64_t n_slots, int32_t d, int32_t n_ids) {
    if (n_slots <= 0 || d <= 0 || n_ids <= 0) return 0;
    return (size_t)embgrad_chunks(n_slots, n_ids) * n_ids * d * sizeof(float);
}

extern "C" int nrms_hier_add_embedding_bwd(int64_t n_slots, int32_t d, int32_t n_ids, const int32_t* id, const int32_t* cnt,
                                           const float* du, float* dtable, void* workspace, size_t workspace_bytes, void* stream) {
    NRMS_REQUIRE(n_slots >= 0 && d > 0 && d <= 1024 && n_ids > 0 && n_ids <= 65535, "hier_add_embedding_bwd: n_slots=%ld d=%d n_ids=%d", (long)n_slots, d, n_ids);
    if (n_slots == 0) return NRMS_OK;
    NRMS_REQUIRE(id && cnt && du && dtable, "hier_add_embedding_bwd: null argument");
    const int nc = embgrad_chunks(n_slots, n_ids);
    const size_t need = (size_t)nc * n_ids * d * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) { set_error("hier_add_embedding_bwd: workspace %zu < required %zu bytes", workspace_bytes, need); return NRMS_EWORKSPACE; }
    const long chunk = ((n_slots + nc - 1) / nc + 63) / 64 * 64;
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("hier_embgrad", s);
    hipLaunchKernelGGL(hier_embgrad_kernel, dim3(n_ids, nc), dim3(256), 0, s, (long)n_slots, chunk, d, id, cnt, du, (float*)workspace);
    const long ne = (long)n_ids * d;
    hipLaunchKernelGGL(hier_embgrad_reduce_kernel, dim3(cdiv(ne, 256)), dim3(256), 0, s, ne, nc, (const float*)workspace, dtable);
    return check_launch("hier_add_embedding_bwd");
}

extern "C" int nrms_hier_match(int32_t B, int32_t C, int32_t H, const int64_t* cand_topic, const int64_t* cand_subtopic,
                               const int32_t* l1_sub, const int32_t* l1_cnt, const int32_t* l2_top, const int32_t* l2_cnt,
                               const int32_t* n_valid, int32_t* sub_slot, float* sub_frac, int32_t* top_slot, float* top_frac,
                               void* stream) {
    NRMS_REQUIRE(B >= 0 && C > 0 && H >= 1 && H <= 64, "hier_match: B=%d C=%d H=%d", B, C, H);
    if (B == 0) return NRMS_OK;
    NRMS_REQUIRE(cand_topic && cand_subtopic && l1_sub && l1_cnt && l2_top && l2_cnt && n_valid && sub_slot && sub_frac && top_slot && top_frac,
                 "hier_match: null argument");
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("hier_match", s);
    hipLaunchKernelGGL(hier_match_kernel, dim3(cdiv((long)B * C, 256)), dim3(256), 0, s, B, C, H, cand_topic, cand_subtopic, l1_sub, l1_cnt,
                       l2_top, l2_cnt, n_valid, sub_slot, sub_frac, top_slot, top_frac);
    return check_launch("hier_match");
}

static int hier_score_args(HierScoreArgs* a, int32_t B, int32_t C, int32_t d, const float* cand, const float* u1, const float* u2,
                           const float* ug, const int32_t* sub_slot, const float* sub_frac, const int32_t* top_slot, const float* top_frac,
                           const uint8_t* mask, float lambda_sub, float lambda_top, const char* who) {
    NRMS_REQUIRE(B >= 0 && C > 0 && d > 0, "%s: B=%d C=%d d=%d", who, B, C, d);
    NRMS_REQUIRE(B == 0 || (cand && u1 && u2 && ug && sub_slot && sub_frac && top_slot && top_frac), "%s: null argument", who);
    a->B = B; a->C = C; a->d = d; a->cand = cand; a->u1 = u1; a->u2 = u2; a->ug = ug;
    a->sub_slot = sub_slot; a->sub_frac = sub_frac; a->top_slot = top_slot; a->top_frac = top_frac; a->mask = mask;
    a->ls = lambda_sub; a->lt = lambda_top; a->lg = 1.0f - lambda_sub - lambda_top;
    return NRMS_OK;
}

extern "C" int nrms_hier_score_fwd(int32_t B, int32_t C, int32_t d, const float* cand, const float* u1, const float* u2,
                                   const float* ug, const int32_t* sub_slot, const float* sub_frac, const int32_t* top_slot,
                                   const float* top_frac, const uint8_t* mask, float lambda_sub, float lambda_top, float* scores,
                                   void* stream) {
    HierScoreArgs a{};
    int rc = hier_score_args(&a, B, C, d, cand, u1, u2, ug, sub_slot, sub_frac, top_slot, top_frac, mask, lambda_sub, lambda_top, "hier_score_fwd");
    if (rc) return rc;
    if (B == 0) return NRMS_OK;
    NRMS_REQUIRE(scores != nullptr, "hier_score_fwd: null scores");
    a.scores = scores;
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("hier_score_fwd", s);
    hipLaunchKernelGGL(hier_score_fwd_kernel, dim3(cdiv((long)B * C, HIER_WPB)), dim3(64 * HIER_WPB), 0, s, a);
    return check_launch("hier_score_fwd");
}

extern "C" int nrms_hier_score_bwd(int32_t B, int32_t C, int32_t d, const float* cand, const float* u1, const float* u2,
                                   const float* ug, const int32_t* sub_slot, const float* sub_frac, const int32_t* top_slot,
                                   const float* top_frac, const uint8_t* mask, float lambda_sub, float lambda_top,
                                   const float* dscores, float* dcand, float* du1, float* du2, float* dug, void* stream) {
    HierScoreArgs a{};
    int rc = hier_score_args(&a, B, C, d, cand, u1, u2, ug, sub_slot, sub_frac, top_slot, top_frac, mask, lambda_sub, lambda_top, "hier_score_bwd");
    if (rc) return rc;
    if (B == 0) return NRMS_OK;
    NRMS_REQUIRE(dscores && dcand && du1 && du2 && dug, "hier_score_bwd: null argument");
    a.dscores = dscores; a.dcand = dcand; a.du1 = du1; a.du2 = du2; a.dug = dug;
    hipStream_t s = (hipStream_t)stream;
    TimingScope ts("hier_score_bwd", s);
    hipLaunchKernelGGL(hier_score_bwd_kernel, dim3(cdiv(B, HIER_WPB)), dim3(64 * HIER_WPB), 0, s, a);
    return check_launch("hier_score_bwd");
}
