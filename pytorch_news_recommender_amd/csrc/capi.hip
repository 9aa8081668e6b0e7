// C ABI of libnrms_hip.so (declared in include/nrms_hip.h): argument validation, the kernel
// sequences of one encoder forward / backward, error text and optional event timing.
#include <stdarg.h>

#include <algorithm>

#include <mutex>
#include <vector>

#include "gemm.h"

namespace nrms {

int launch_attention(bool bwd, int n_seq, int S, int d, int h, const float* qkv, float* ctx, const Dropout& drop,
                     const float* dctx, float* dqkv, const uint8_t* mask, const int64_t* ids, const float* bias_hm,
                     const int* pos, float* padsum, float* dbias, hipStream_t stream, const Dropout* pdrop = nullptr,
                     bool split = false);
size_t attention_padsum_floats();
int launch_addattn_fwd(int n_seq, int S, int d, int q, const float* ctx, const float* w_add, const float* b_add,
                       const float* q_vec, float* T, float* wout, float* out, const uint8_t* mask, int npass,
                       void* wplanes, hipStream_t stream);
int addattn_bwd_rows_waves(int n_seq);
int launch_addattn_bwd_rows(int n_seq, int S, int d, int q, const float* ctx, const float* dout, const float* w,
                            const float* T, float* ds, float* dq_partial, float* dq, const uint8_t* mask,
                            hipStream_t stream);

// wide.hip: shape-general fp32 kernels (nrms_naml's widths, dropout on the attention probabilities)
int launch_attention_wide(bool bwd, int n_seq, int S, int d, int h, const float* qkv, float* ctx, const Dropout& pdrop,
                          const float* dctx, float* dqkv, const uint8_t* mask, hipStream_t stream);
int launch_addattn_rows_fwd_wide(int n_seq, int S, int d, int q, float* T, const float* q_vec, const float* ctx,
                                 const uint8_t* mask, float* wout, float* out, hipStream_t stream);
int launch_addattn_rows_bwd_wide(int n_seq, int S, int d, int q, const float* ctx, const float* dout, const float* w,
                                 const float* T, float* ds, float* dq_partial, float* dq, const uint8_t* mask,
                                 hipStream_t stream);
int launch_layernorm_fwd(long n_rows, int d, const float* x, const float* gamma, const float* beta, float eps, float* y,
                         float* stats, hipStream_t stream);
size_t layernorm_bwd_workspace_floats(int d);
int launch_layernorm_bwd(long n_rows, int d, const float* x, const float* gamma, const float* stats, const float* dy,
                         float* dx, float* dgamma_dbeta, float* workspace, hipStream_t stream);
struct FeatArgs {
    long n;
    int dt, dc, n_cat, n_sub;
    const float *title, *abst, *cat_table, *sub_table;
    const int64_t *categ, *subcateg;
    Dropout drop;
    float* out;
    const float* dout;
    float *d_title, *d_abst, *d_cat_table, *d_sub_table;
};
int launch_features_fwd(const FeatArgs& a, hipStream_t stream);
int launch_features_bwd(const FeatArgs& a, hipStream_t stream);

// ---- error text -----------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return NRMS_ELAUNCH;
    }
    return NRMS_OK;
}

// ---- event timing ---------------------------------------------------------------------
struct TimedLaunch { std::string name; hipEvent_t start, stop; };
static std::mutex g_tmu;
static bool g_timing = false;
static std::vector<TimedLaunch> g_launches;

TimingScope::TimingScope(const char* name, hipStream_t s) : active(false), slot(-1), stream(s) {
    if (!g_timing) return;
    std::lock_guard<std::mutex> lk(g_tmu);
    TimedLaunch t;
    t.name = name;
    if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess) return;
    (void)hipEventRecord(t.start, s);
    g_launches.push_back(t);
    slot = (int)g_launches.size() - 1;
    active = true;
}

TimingScope::~TimingScope() {
    if (!active) return;
    std::lock_guard<std::mutex> lk(g_tmu);
    (void)hipEventRecord(g_launches[slot].stop, stream);
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- encoder --------------------------------------------------------------------------
// the MFMA attention kernels take even d_k <= 64; anything else goes to wide.hip
static bool wide_attention(const nrms_encoder_desc* d) {
    const int dk = d->n_heads > 0 ? d->d_model / d->n_heads : 0;
    return dk > 64 || (dk & 1) != 0;
}
// the fused additive-attention forward holds q_dim <= 256 accumulator columns, the row backward d_model <= 512
static bool wide_additive(const nrms_encoder_desc* d) { return d->q_dim > 256 || d->d_model > 512; }

static int validate_desc(const nrms_encoder_desc* d, const char* who) {
    NRMS_REQUIRE(d != nullptr, "%s: null desc", who);
    NRMS_REQUIRE(d->n_seq >= 0, "%s: n_seq=%d", who, d->n_seq);
    NRMS_REQUIRE(d->seq_len >= 1 && d->seq_len <= 64, "%s: seq_len=%d outside 1..64", who, d->seq_len);
    NRMS_REQUIRE(d->d_model > 0 && (d->d_model & 3) == 0 && d->d_model <= 1024,
                 "%s: d_model=%d must be a positive multiple of 4, <= 1024", who, d->d_model);
    NRMS_REQUIRE(d->n_heads > 0 && d->d_model % d->n_heads == 0, "%s: d_model %% n_heads != 0", who);
    const int dk = d->d_model / d->n_heads;
    NRMS_REQUIRE(dk <= 128, "%s: d_k=%d must be <= 128", who, dk);
    NRMS_REQUIRE(d->q_dim > 0 && (d->q_dim & 3) == 0 && d->q_dim <= 512, "%s: q_dim=%d must be a multiple of 4 <= 512",
                 who, d->q_dim);
    NRMS_REQUIRE(d->p_drop_attn >= 0.f && d->p_drop_attn < 1.f, "%s: p_drop_attn must be in [0,1)", who);
    if (wide_attention(d) && d->precision != NRMS_PRECISION_FP16)       // (the fused fp16 kernels have their own shape rules, below)
        NRMS_REQUIRE(!(d->vocab > 0 && (d->flags & NRMS_FLAG_PAD_ROW_ZERO)) && d->p_drop_ctx == 0.f,
                     "%s: d_k > 64 or odd d_k (the shape-general attention) supports neither NRMS_FLAG_PAD_ROW_ZERO nor "
                     "p_drop_ctx", who);
    // (p_drop_attn with NRMS_FLAG_PAD_ROW_ZERO: the token compaction stays exact; the closed form of an all-padding
    //  sequence gains the per-query kept-key factors, see allpad_keep_factor in attention.hip)
    NRMS_REQUIRE(d->vocab >= 0, "%s: vocab=%d", who, d->vocab);
    NRMS_REQUIRE(d->p_drop_embed >= 0.f && d->p_drop_embed < 1.f && d->p_drop_ctx >= 0.f && d->p_drop_ctx < 1.f,
                 "%s: dropout probabilities must be in [0,1)", who);
    NRMS_REQUIRE(d->vocab > 0 || d->p_drop_embed == 0.f, "%s: embedding dropout needs the news encoder (vocab>0)", who);
    NRMS_REQUIRE(d->precision >= NRMS_PRECISION_FP32 && d->precision <= NRMS_PRECISION_FP16,
                 "%s: unsupported precision %d", who, d->precision);
    if (d->precision == NRMS_PRECISION_FP16) {
        const char* why = nullptr;
        if (d->use_output_proj) {
            // wide heads + W_O (nrms_v1's news encoder): fused16_v1.hip, padding-skipping path only
            NRMS_REQUIRE(fused16v1_supported(d->seq_len, d->d_model, d->n_heads, d->q_dim, &why),
                         "%s: precision fp16 with the output projection needs %s (use bf16x3 for this shape)", who, why);
            NRMS_REQUIRE(d->vocab > 0 && (d->flags & NRMS_FLAG_PAD_ROW_ZERO) != 0 && d->p_drop_embed == 0.f,
                         "%s: precision fp16 with the output projection covers the news encoder with NRMS_FLAG_PAD_ROW_ZERO "
                         "and no embedding dropout (use bf16x3)", who);
        } else {
            NRMS_REQUIRE(fused16_supported(d->seq_len, d->d_model, d->n_heads, d->q_dim, &why),
                         "%s: precision fp16 needs %s (use bf16x3 for this shape)", who, why);
        }
        NRMS_REQUIRE(d->mask_mode == 0 && d->p_drop_attn == 0.f,
                     "%s: precision fp16 supports neither masks nor p_drop_attn (use bf16x3)", who);
    }
    NRMS_REQUIRE((d->mask_mode & ~3) == 0, "%s: mask_mode=%d", who, d->mask_mode);
    NRMS_REQUIRE((d->flags & ~(NRMS_FLAG_PAD_ROW_ZERO | NRMS_FLAG_DEFER_WQKV | NRMS_FLAG_FWD_SCRATCH_KEPT | NRMS_FLAG_FUSED_SEQ64)) == 0,
                 "%s: unknown flags 0x%x", who, d->flags);
    if (d->flags & NRMS_FLAG_FUSED_SEQ64) {
        const char* why = nullptr;
        NRMS_REQUIRE(d->vocab == 0 && d->precision == NRMS_PRECISION_BF16X3 && !d->use_output_proj && d->mask_mode == 0 &&
                     d->p_drop_ctx == 0.f && d->p_drop_attn == 0.f,
                     "%s: NRMS_FLAG_FUSED_SEQ64 covers the user encoder (vocab == 0) in bf16x3 without output projection, mask or dropout", who);
        NRMS_REQUIRE(user64_supported(d->seq_len, d->d_model, d->n_heads, d->q_dim, &why), "%s: NRMS_FLAG_FUSED_SEQ64 needs %s", who, why);
    }
    NRMS_REQUIRE((long)d->n_seq * d->seq_len < (1L << 31), "%s: n_seq*seq_len overflows int32", who);
    if (d->seq_index != nullptr)
        NRMS_REQUIRE(d->precision != NRMS_PRECISION_FP16 && (d->flags & NRMS_FLAG_FUSED_SEQ64) == 0 && !wide_attention(d) &&
                     d->p_drop_embed == 0.f && d->p_drop_ctx == 0.f,
                     "%s: seq_index renumbers the counters of the attention-probability dropout only (fp32 / bf16x3 / bf16, d_k <= 64 even, "
                     "no embedding or context dropout)", who);
    return NRMS_OK;
}

// NT GEMM in the precision the descriptor asks for
static int nt_gemm(const nrms_encoder_desc* d, int amode, int emode, const NTArgs& g, void* wplanes, hipStream_t s,
                   const char* name) {
    if (d->precision == NRMS_PRECISION_FP32) return launch_gemm_nt(amode, emode, g, s, name);
    return launch_gemm_nt_bf16(amode, emode, d->precision == NRMS_PRECISION_BF16X3 ? 3 : 1, g, wplanes, s, name);
}

static int tn_gemm(const nrms_encoder_desc* d, const TNArgs& t, hipStream_t s, const char* name) {
    if (d->precision == NRMS_PRECISION_FP32) return launch_gemm_tn(t, s, name);
    return launch_gemm_tn_bf16(d->precision == NRMS_PRECISION_BF16X3 ? 3 : 1, t, s, name);
}

static size_t wplane_bytes(const nrms_encoder_desc* d) {
    if (d->precision == NRMS_PRECISION_FP32 || d->precision == NRMS_PRECISION_FP16) return 0;
    const int dm = d->d_model, q = d->q_dim;
    size_t m = gemm_nt_bf16_wplane_bytes(3 * dm, dm);                    // QKV
    const size_t a = gemm_nt_bf16_wplane_bytes(dm, 3 * dm);             // dX
    const size_t b = gemm_nt_bf16_wplane_bytes(dm, q);                  // dctx
    if (a > m) m = a;
    if (b > m) m = b;
    return align_up(m, 256);
}

struct BwdWorkspace {
    size_t dctx, dqkv, dattn, ds, wqkv_t, wadd_t, wo_t, tn_partial, dq_partial, wplanes, live, pos, n_live, cscr, padsum, sscr, u64planes, total;   // byte offsets
};

static BwdWorkspace bwd_layout(const nrms_encoder_desc* d) {
    const size_t M = (size_t)d->n_seq * d->seq_len, dm = d->d_model, q = d->q_dim;
    BwdWorkspace w;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats * sizeof(float), 256); return o; };
    w.dctx = take(M * dm);
    w.dqkv = take(M * 3 * dm);
    w.dattn = take(d->use_output_proj ? M * dm : 0);
    w.ds = take(M);
    w.wqkv_t = take(3 * dm * dm);
    w.wadd_t = take(q * dm);
    w.wo_t = take(d->use_output_proj ? dm * dm : 0);
    const size_t p1 = gemm_tn_workspace_floats((int)M, (int)(3 * dm), (int)dm, nullptr);
    const size_t p2 = gemm_tn_workspace_floats((int)M, (int)q, (int)dm, nullptr);
    const size_t p3 = d->use_output_proj ? gemm_tn_workspace_floats((int)M, (int)dm, (int)dm, nullptr) : 0;
    w.tn_partial = take(p1 > p2 ? (p1 > p3 ? p1 : p3) : (p2 > p3 ? p2 : p3));
    const bool fused64 = (d->flags & NRMS_FLAG_FUSED_SEQ64) != 0;
    w.dq_partial = take(std::max((size_t)(wide_additive(d) ? d->n_seq : addattn_bwd_rows_waves(d->n_seq)) * q,
                                 fused64 ? user64_dq_partial_floats(d->n_seq) : (size_t)0));
    w.wplanes = take(wplane_bytes(d) / sizeof(float));
    const bool pz = d->vocab > 0 && (d->flags & NRMS_FLAG_PAD_ROW_ZERO) != 0;
    w.live = take(d->vocab > 0 ? M : 0);          // int32 positions of the non-padding tokens (news encoder)
    w.pos = take(pz ? M : 0);                     // inverse map
    w.n_live = take(d->vocab > 0 ? 64 : 0);
    w.cscr = take(d->vocab > 0 ? compact_scratch_ints((long)M) : 0);
    w.padsum = take(pz ? attention_padsum_floats() : 0);
    w.sscr = take(d->vocab > 0 ? scatter_grouped_scratch_ints((long)M, d->vocab) : 0);
    w.u64planes = take(fused64 ? user64_bwd_planes_bytes(d->n_heads) / sizeof(float) + 1 : 0);
    w.total = off;
    return w;
}

}  // namespace nrms

using namespace nrms;

// forward scratch: head-major copies of W_qkv / b_qkv (HeadPerm, common.h), then the bf16 weight planes
struct FwdScratch { size_t wq, bq, planes, live, n_live, cscr, total; };
static bool skip_pad_rows(const nrms_encoder_desc* d) { return d->vocab > 0 && (d->flags & NRMS_FLAG_PAD_ROW_ZERO) != 0; }
static FwdScratch fwd_scratch(const nrms_encoder_desc* d) {
    FwdScratch f;
    const size_t dm = (size_t)d->d_model;
    f.wq = 0;
    f.bq = align_up(3 * dm * dm * sizeof(float), 256);
    f.planes = f.bq + align_up(3 * dm * sizeof(float), 256);
    f.live = f.planes + wplane_bytes(d);                 // wplane_bytes is 256-aligned
    f.n_live = f.live + (skip_pad_rows(d) ? align_up((size_t)d->n_seq * d->seq_len * sizeof(int), 256) : 0);
    f.cscr = f.n_live + (skip_pad_rows(d) ? 256 : 0);
    f.total = f.cscr + (skip_pad_rows(d) ? align_up(compact_scratch_ints((long)d->n_seq * d->seq_len) * sizeof(int), 256) : 0);
    if (d->flags & NRMS_FLAG_FUSED_SEQ64) f.total = std::max(f.total, align_up(user64_fwd_planes_bytes(d->n_heads), 256));
    return f;
}

// ---- fp16 mode: planes | live | pos | n_live | compaction scratch
struct Fwd16Scratch { size_t planes, live, pos, n_live, cscr, order, order_cnt, total; };
static Fwd16Scratch fwd16_scratch(const nrms_encoder_desc* d) {
    Fwd16Scratch f;
    const size_t M = (size_t)d->n_seq * d->seq_len;
    const bool gather = d->vocab > 0;
    f.planes = 0;
    f.live = align_up(d->use_output_proj ? fused16v1_planes_bytes(d->d_model, d->n_heads, d->q_dim)
                                         : fused16_layout(d->d_model, d->n_heads, d->q_dim).total, 256);
    f.pos = f.live + (gather ? align_up(M * sizeof(int), 256) : 0);
    f.n_live = f.pos + (gather ? align_up(M * sizeof(int), 256) : 0);
    f.cscr = f.n_live + (gather ? 256 : 0);
    f.order = f.cscr + (gather ? align_up(compact_scratch_ints((long)M) * sizeof(int), 256) : 0);
    f.order_cnt = f.order + (gather ? align_up((size_t)3 * d->n_seq * sizeof(int), 256) : 0);
    f.total = f.order_cnt + (gather ? align_up(title_order_cnt_ints(d->n_seq) * sizeof(int), 256) : 0);
    return f;
}

// ---- NRMS_FLAG_FWD_SCRATCH_KEPT is a promise of the caller; the library checks it on the HOST.  Every nrms_encoder_fwd forgets
// what it knew about the scratch buffer it is handed, and an fp16 news-encoder forward that builds token / title lists in it
// records (scratch, ids, shape).  A backward with the flag uses the lists only if that record still matches its own arguments;
// otherwise it rebuilds them from the ids, as without the flag -- a caller that reused the scratch for another forward, or
// changed n_seq or the ids, gets correct gradients instead of out-of-bounds list entries.  (Memory the caller overwrote by other
// means is beyond what a host-side record can see.)  One mutex-protected table per process, at most 64 entries.
struct KeptRec { const void* scratch; const void* ids; int n_seq, seq_len, vocab, d_model, n_heads, use_wo; };
static std::mutex g_kept_mu;
static std::vector<KeptRec> g_kept;
static void kept_forget(const void* scratch) {
    if (scratch == nullptr) return;
    std::lock_guard<std::mutex> lk(g_kept_mu);
    for (size_t i = 0; i < g_kept.size(); ++i)
        if (g_kept[i].scratch == scratch) { g_kept[i] = g_kept.back(); g_kept.pop_back(); break; }
}
static void kept_record(const nrms_encoder_desc* d, const void* scratch, const void* ids) {
    std::lock_guard<std::mutex> lk(g_kept_mu);
    if (g_kept.size() >= 64) g_kept.erase(g_kept.begin());
    g_kept.push_back(KeptRec{scratch, ids, d->n_seq, d->seq_len, d->vocab, d->d_model, d->n_heads, d->use_output_proj});
}
static bool kept_matches(const nrms_encoder_desc* d, const void* scratch, const void* ids) {
    std::lock_guard<std::mutex> lk(g_kept_mu);
    for (const KeptRec& r : g_kept)
        if (r.scratch == scratch)
            return r.ids == ids && r.n_seq == d->n_seq && r.seq_len == d->seq_len && r.vocab == d->vocab && r.d_model == d->d_model &&
                   r.n_heads == d->n_heads && r.use_wo == d->use_output_proj;
    return false;
}

static int encoder_fwd16(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int64_t* ids, const float* x,
                         const nrms_encoder_acts* acts, float* out, hipStream_t s) {
    const bool gather = desc->vocab > 0;
    NRMS_REQUIRE(acts->x && acts->ctx && acts->scratch, "encoder_fwd(fp16): acts.x, acts.ctx and acts.scratch are required");
    const int S = desc->seq_len, d = desc->d_model, h = desc->n_heads, q = desc->q_dim;
    const long M = (long)desc->n_seq * S;
    const Fwd16Scratch fs = fwd16_scratch(desc);
    const Fused16Layout L = fused16_layout(d, h, q);
    char* base = (char*)acts->scratch;
    const bool v1 = desc->use_output_proj != 0;
    NRMS_REQUIRE(!v1 || (w->w_o && w->b_o && acts->attn), "encoder_fwd(fp16): use_output_proj needs w_o, b_o and acts.attn");
    // The weight planes (weights only) and the title lists (ids only) are built on helper stream 1 while the caller's stream
    // compacts the token rows and gathers the embeddings (event 5 forks, event 4 joins in front of the fused kernel)
    SideSet* ss = (gather && skip_pad_rows(desc)) ? side_streams_for(s) : nullptr;
    hipStream_t s_prep = ss != nullptr ? ss->s[1] : s;
    int rc = fused_bwd16_join(s);                  // a deferred join of an earlier backward on this stream uses the same events
    if (rc) return rc;
    if (ss != nullptr) {
        rc = side_order(ss, 5, s, s_prep, "encoder_fwd(fp16)");
        if (rc) return rc;
    }
    bool forked = ss != nullptr;
    auto join_prep = [&]() -> int {                 // (also on the error paths: the caller may reuse the scratch)
        if (!forked) return NRMS_OK;
        forked = false;
        if (hipEventRecord(ss->ev[4], s_prep) != hipSuccess || hipStreamWaitEvent(s, ss->ev[4], 0) != hipSuccess) {
            set_error("encoder_fwd(fp16): joining the helper stream failed");
            return NRMS_ELAUNCH;
        }
        return NRMS_OK;
    };
    rc = v1 ? launch_prep16v1(d, h, q, w->w_qkv, w->b_qkv, w->w_o, w->b_o, w->w_add, w->b_add, w->q_vec, base + fs.planes, s_prep)
            : launch_prep16(d, h, q, w->w_qkv, w->b_qkv, w->w_add, w->b_add, w->q_vec, base + fs.planes, s_prep);
    if (rc) { (void)join_prep(); return rc; }
    Fused16Fwd f{};
    f.n_seq = desc->n_seq; f.S = S; f.d = d; f.h = h; f.q = q;
    f.planes = base + fs.planes; f.x16 = acts->x; f.ctx16 = acts->ctx; f.t16 = acts->t; f.w = acts->w; f.out = out;
    f.drop = make_dropout(desc->seed, desc->p_drop_ctx);
    if (gather) {
        const Dropout drop_e = make_dropout(desc->seed, desc->p_drop_embed);
        if (skip_pad_rows(desc)) {
            int* live = (int*)(base + fs.live);
            int* pos = (int*)(base + fs.pos);
            int* n_live = (int*)(base + fs.n_live);
            int* order = (int*)(base + fs.order);
            int* order_cnt = (int*)(base + fs.order_cnt);
            rc = launch_title_order(desc->n_seq, S, ids, order, order_cnt, s_prep, 3);
            if (rc) { (void)join_prep(); return rc; }
            rc = launch_compact_live_rows(M, ids, live, pos, n_live, (int*)(base + fs.cscr), s);
            if (rc == NRMS_OK) rc = launch_gather16(M, d, L.KP, ids, live, n_live, w->table, drop_e, acts->x, s);
            const int rj = join_prep();
            if (rc) return rc;
            if (rj) return rj;
            f.pos = pos;
            f.n_rows = n_live;
            f.ids = ids;
            f.order = order; f.order_cnt = order_cnt;
        } else {
            rc = launch_gather16(M, d, L.KP, ids, nullptr, nullptr, w->table, drop_e, acts->x, s);
            if (rc) return rc;
        }
    } else {
        rc = launch_cast16(M, d, L.KP, x, acts->x, s);
        if (rc) return rc;
    }
    rc = v1 ? launch_fused_fwd16v1(f, h, acts->attn, s) : launch_fused_fwd16(f, s);
    if (rc == NRMS_OK && gather && skip_pad_rows(desc)) kept_record(desc, acts->scratch, ids);
    return rc;
}

// ---- fp16 backward workspace: fused16_bwd_layout | live | pos | n_live | compaction scratch | order | order_cnt |
//      compact dX (news encoder) | scatter scratch
struct Bwd16Layout { size_t fused, live, pos, n_live, cscr, order, order_cnt, dxc, sscr, total; };
static Bwd16Layout bwd16_layout(const nrms_encoder_desc* d) {
    Bwd16Layout L;
    const size_t M = (size_t)d->n_seq * d->seq_len;
    const bool gather = d->vocab > 0;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.fused = take(d->use_output_proj ? fused16v1_bwd_bytes((long)M, d->n_seq, d->n_heads) : fused16_bwd_layout((long)M, d->n_seq).total);
    L.live = take(gather ? M * sizeof(int) : 0);
    L.pos = take(gather ? M * sizeof(int) : 0);
    L.n_live = take(gather ? 256 : 0);
    L.cscr = take(gather ? compact_scratch_ints((long)M) * sizeof(int) : 0);
    L.order = take(gather ? (size_t)3 * d->n_seq * sizeof(int) : 0);
    L.order_cnt = take(gather ? title_order_cnt_ints(d->n_seq) * sizeof(int) : 0);
    L.dxc = take(gather ? M * std::max((size_t)d->d_model * sizeof(float), (size_t)NRMS_FP16_KP * 2) : 0);   // fp32 [M][d] or fp16 [M][320]
    L.sscr = take(gather ? scatter_grouped_scratch_ints((long)M, d->vocab) * sizeof(int) : 0);
    L.total = off;
    return L;
}

static int encoder_bwd16(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int64_t* ids, const float* x,
                         const nrms_encoder_acts* acts, const float* dout, const nrms_encoder_grads* grads, float* dx,
                         void* workspace, size_t workspace_bytes, hipStream_t s) {
    NRMS_REQUIRE(w && acts && dout && grads && workspace, "encoder_bwd(fp16): null argument");
    const bool v1 = desc->use_output_proj != 0;
    NRMS_REQUIRE(!v1 || (w->w_o && acts->attn && grads->w_o && grads->b_o),
                 "encoder_bwd(fp16): use_output_proj needs w_o, acts.attn, grads.w_o, grads.b_o");
    NRMS_REQUIRE(acts->x && acts->ctx && acts->t && acts->w, "encoder_bwd(fp16): acts.x, ctx, t, w are required");
    NRMS_REQUIRE(grads->w_qkv && grads->b_qkv && grads->w_add && grads->b_add && grads->q_vec, "encoder_bwd(fp16): null gradient buffer");
    const bool gather = desc->vocab > 0;
    NRMS_REQUIRE(gather ? (ids != nullptr && grads->table != nullptr) : (dx != nullptr), "encoder_bwd(fp16): %s missing",
                 gather ? "ids / grads.table" : "dx");
    const Bwd16Layout L = bwd16_layout(desc);
    if (workspace_bytes < L.total) { set_error("encoder_bwd(fp16): workspace %zu < required %zu bytes", workspace_bytes, L.total); return NRMS_EWORKSPACE; }
    NRMS_REQUIRE(((uintptr_t)workspace & 255) == 0, "encoder_bwd(fp16): workspace must be 256-byte aligned");
    if (desc->n_seq == 0) return NRMS_OK;
    const int S = desc->seq_len, d = desc->d_model;
    const long M = (long)desc->n_seq * S;
    char* base = (char*)workspace;
    Fused16Bwd f{};
    f.n_seq = desc->n_seq; f.S = S; f.d = d; f.h = desc->n_heads; f.q = desc->q_dim;
    f.workspace = base + L.fused;
    f.w_qkv = w->w_qkv; f.b_qkv = w->b_qkv; f.w_add = w->w_add; f.q_vec = w->q_vec;
    f.x16 = acts->x; f.ctx16 = acts->ctx; f.t16 = acts->t; f.w = acts->w; f.dout = dout;
    f.loss_scale = desc->loss_scale;           // <= 0: device-side, from max |dout|
    f.drop = make_dropout(desc->seed, desc->p_drop_ctx);
    f.dw_qkv = grads->w_qkv; f.db_qkv = grads->b_qkv; f.dw_add = grads->w_add; f.db_add = grads->b_add; f.dq_vec = grads->q_vec;
    if (v1) { f.w_o = w->w_o; f.attn16 = acts->attn; f.dw_o = grads->w_o; f.db_o = grads->b_o; }
    int rc;
    int* live = (int*)(base + L.live);
    int* n_live = (int*)(base + L.n_live);
    if (gather) {
        const bool kept = (desc->flags & NRMS_FLAG_FWD_SCRATCH_KEPT) != 0 && skip_pad_rows(desc) && acts->scratch != nullptr &&
                          kept_matches(desc, acts->scratch, ids);          // (a stale promise: rebuild, below)
        if (kept) {
            // the lists of this step's forward, where encoder_fwd16 left them
            const Fwd16Scratch fs = fwd16_scratch(desc);
            char* fb = (char*)acts->scratch;
            live = (int*)(fb + fs.live);
            n_live = (int*)(fb + fs.n_live);
            f.pos = (int*)(fb + fs.pos); f.n_rows_dev = n_live; f.ids = ids;
            f.order = (int*)(fb + fs.order); f.order_cnt = (int*)(fb + fs.order_cnt);
        } else {
            // the forward's scratch may have been reused by later forward calls: rebuild the lists
            int* pos = (int*)(base + L.pos);
            rc = launch_compact_live_rows(M, ids, live, pos, n_live, (int*)(base + L.cscr), s);
            if (rc) return rc;
            if (skip_pad_rows(desc)) {
                int* order = (int*)(base + L.order);
                int* order_cnt = (int*)(base + L.order_cnt);
                rc = launch_title_order(desc->n_seq, S, ids, order, order_cnt, s, 3);
                if (rc) return rc;
                f.pos = pos; f.n_rows_dev = n_live; f.ids = ids; f.order = order; f.order_cnt = order_cnt;
            }
        }
        f.dx = (float*)(base + L.dxc);
        // compact path: dX travels to the grouped scatter as fp16 (still loss-scaled)
        f.dx_fp16 = skip_pad_rows(desc);
    } else {
        f.dx = dx;
    }
    // the grouped scatter's lists depend on the ids only: built on helper stream 1 beside the fused kernels (event 5 forks,
    // event 4 joins in front of the scatter -- and on every error path, so the caller may reuse the workspace)
    SideSet* ss_prep = (gather && skip_pad_rows(desc)) ? side_streams_for(s) : nullptr;
    bool prepared = false;
    if (ss_prep != nullptr) {
        rc = side_order(ss_prep, 5, s, ss_prep->s[1], "encoder_bwd(fp16)");
        if (rc) return rc;
        rc = launch_scatter_prepare(M, desc->vocab, ids, live, n_live, (int*)(base + L.sscr), ss_prep->s[1]);
        if (hipEventRecord(ss_prep->ev[4], ss_prep->s[1]) != hipSuccess) { set_error("encoder_bwd(fp16): hipEventRecord failed"); return NRMS_ELAUNCH; }
        if (rc) { (void)hipStreamWaitEvent(s, ss_prep->ev[4], 0); return rc; }
        prepared = true;
    }
    auto join_prepare = [&]() -> int {
        if (!prepared) return NRMS_OK;
        if (hipStreamWaitEvent(s, ss_prep->ev[4], 0) != hipSuccess) { set_error("encoder_bwd(fp16): hipStreamWaitEvent failed"); return NRMS_ELAUNCH; }
        return NRMS_OK;
    };
    const float* sc_dev = nullptr;
    f.sc_out = &sc_dev;
    // NRMS_FLAG_DEFER_WQKV: the two weight-gradient GEMMs stay on their helper streams; dX and the table gradient below
    // are complete in stream order, nrms_encoder_bwd_wqkv joins the rest (a data-parallel caller starts the table
    // all-reduce in between)
    f.defer_join = (desc->flags & NRMS_FLAG_DEFER_WQKV) != 0;
    rc = v1 ? launch_fused_bwd16v1(f, s) : launch_fused_bwd16(f, s);
    {
        const int rj = join_prepare();
        if (rc) return rc;
        if (rj) return rj;
    }
    if (gather) {
        const Dropout drop_e = make_dropout(desc->seed, desc->p_drop_embed);
        if (skip_pad_rows(desc))
            rc = launch_scatter_grouped(M, desc->vocab, d, ids, live, n_live, f.dx, drop_e, grads->table, (int*)(base + L.sscr), s,
                                        f.dx_fp16, NRMS_FP16_KP, sc_dev, prepared);
        else       // dense rows (one per token): the atomic scatter walks the live list over the dense rows
            rc = launch_scatter_dense_rows(M, d, ids, live, n_live, f.dx, drop_e, grads->table, s);
    }
    return rc;
}

extern "C" int nrms_encoder_fwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int64_t* ids,
                                const float* x, const uint8_t* mask, const nrms_encoder_acts* acts, float* out,
                                void* stream) {
    int rc = validate_desc(desc, "encoder_fwd");
    if (rc) return rc;
    if (acts != nullptr) kept_forget(acts->scratch);          // whatever lists an earlier forward left in this scratch are gone
    if (desc->precision == NRMS_PRECISION_FP16) {
        NRMS_REQUIRE(w && acts && out, "encoder_fwd: null argument");
        NRMS_REQUIRE(w->w_qkv && w->b_qkv && w->w_add && w->b_add && w->q_vec, "encoder_fwd: null weight");
        NRMS_REQUIRE(desc->vocab > 0 ? (ids != nullptr && w->table != nullptr) : (x != nullptr), "encoder_fwd: input missing");
        if (desc->n_seq == 0) return NRMS_OK;
        return encoder_fwd16(desc, w, ids, x, acts, out, (hipStream_t)stream);
    }
    NRMS_REQUIRE(w && acts && out, "encoder_fwd: null argument");
    NRMS_REQUIRE(w->w_qkv && w->b_qkv && w->w_add && w->b_add && w->q_vec, "encoder_fwd: null weight");
    if (desc->flags & NRMS_FLAG_FUSED_SEQ64) {
        // the user encoder as one kernel (csrc/user64.hip); acts.qkv holds operand fragments (nrms_encoder_fused_qkv_bytes)
        NRMS_REQUIRE(x != nullptr && acts->qkv != nullptr && acts->scratch != nullptr, "encoder_fwd(fused): x, acts.qkv and acts.scratch are required");
        const bool train = acts->ctx != nullptr && acts->t != nullptr && acts->w != nullptr;
        if (desc->n_seq == 0) return NRMS_OK;
        rc = fused_bwd16_join((hipStream_t)stream);
        if (rc) return rc;
        return launch_user64_fwd(desc->n_seq, desc->seq_len, desc->d_model, desc->n_heads, desc->q_dim, x, w->w_qkv, w->b_qkv, w->w_add,
                                 w->b_add, w->q_vec, acts->scratch, acts->ctx, acts->t, acts->w, acts->qkv, out, train, (hipStream_t)stream);
    }
    NRMS_REQUIRE(acts->qkv && acts->ctx, "encoder_fwd: acts.qkv / acts.ctx are required");
    const bool gather = desc->vocab > 0, wo = desc->use_output_proj != 0;
    NRMS_REQUIRE(gather ? (ids != nullptr && w->table != nullptr) : (x != nullptr),
                 "encoder_fwd: %s input missing", gather ? "ids/table" : "x");
    NRMS_REQUIRE(!wo || (w->w_o && w->b_o && acts->attn), "encoder_fwd: use_output_proj needs w_o, b_o and acts.attn");
    NRMS_REQUIRE(desc->mask_mode == 0 || mask != nullptr, "encoder_fwd: mask_mode=%d but mask is null", desc->mask_mode);
    if (desc->n_seq == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int S = desc->seq_len, d = desc->d_model, q = desc->q_dim, M = desc->n_seq * S;
    const Dropout drop_e = make_dropout(desc->seed, desc->p_drop_embed);
    const Dropout drop_c = make_dropout(desc->seed, desc->p_drop_ctx);
    const Dropout no_drop = make_dropout(0, 0.f);
    const uint8_t* amask = (desc->mask_mode & 1) ? mask : nullptr;
    const uint8_t* pmask = (desc->mask_mode & 2) ? mask : nullptr;

    NRMS_REQUIRE(acts->scratch != nullptr, "encoder_fwd: acts.scratch (nrms_encoder_fwd_scratch_bytes) is required");
    const FwdScratch fs = fwd_scratch(desc);
    float* wq_hm = (float*)((char*)acts->scratch + fs.wq);
    float* bq_hm = (float*)((char*)acts->scratch + fs.bq);
    void* wplanes = (char*)acts->scratch + fs.planes;
    const HeadPerm perm{d / desc->n_heads, desc->n_heads};
    rc = launch_permute_rows(w->w_qkv, w->b_qkv, wq_hm, bq_hm, 3 * d, d, perm, s);
    if (rc) return rc;

    NTArgs g{};
    g.M = M; g.N = 3 * d; g.K = d; g.rows_per_tile = NT_BM;
    g.W = wq_hm; g.bias = bq_hm; g.C = acts->qkv; g.ldc = 3 * d;
    const float* xin = x;
    if (gather) {
        NRMS_REQUIRE(acts->x != nullptr, "encoder_fwd: acts.x (gathered embeddings) is required for the news encoder");
        if (skip_pad_rows(desc)) {
            // NRMS_FLAG_PAD_ROW_ZERO: a padding token's embedding is exactly zero, so its Q|K|V row is exactly
            // the bias.  acts.x holds ONLY the non-padding tokens (compact, ascending token order), the GEMM
            // runs on those rows and scatters its C rows back to the token positions; the padding rows of
            // qkv are filled with the bias.
            int* live = (int*)((char*)acts->scratch + fs.live);
            int* n_live = (int*)((char*)acts->scratch + fs.n_live);
            int* cscr = (int*)((char*)acts->scratch + fs.cscr);
            rc = launch_compact_live_rows((long)M, ids, live, nullptr, n_live, cscr, s);
            if (rc) return rc;
            rc = launch_gather_dropout_compact((long)M, d, ids, live, n_live, w->table, drop_e, acts->x, s);
            if (rc) return rc;
            rc = launch_fill_pad_rows((long)desc->n_seq, S, 3 * d, ids, bq_hm, acts->qkv, amask == nullptr, s);
            if (rc) return rc;
            g.c_rows = live; g.m_dev = n_live;
        } else {
            rc = launch_gather_dropout((long)M, d, ids, w->table, drop_e, acts->x, s);
            if (rc) return rc;
        }
        xin = acts->x;
    }
    g.A = xin; g.lda = d;
    rc = nt_gemm(desc, A_PLAIN, E_STORE, g, wplanes, s, "qkv_proj_fwd");
    if (rc) return rc;
    // v0: the attention kernel writes ctx through the context dropout.  v1: it writes the raw head
    // concatenation, the output projection follows and carries the dropout in its epilogue.
    Dropout pdrop_attn = make_dropout(desc->seed, desc->p_drop_attn);
    pdrop_attn.seq_index = desc->seq_index; pdrop_attn.seq_h = desc->n_heads;
    if (wide_attention(desc))
        rc = launch_attention_wide(false, desc->n_seq, S, d, desc->n_heads, acts->qkv, wo ? acts->attn : acts->ctx,
                                   pdrop_attn, nullptr, nullptr, amask, s);
    else
        rc = launch_attention(false, desc->n_seq, S, d, desc->n_heads, acts->qkv, wo ? acts->attn : acts->ctx,
                              wo ? no_drop : drop_c, nullptr, nullptr, amask,
                              (gather && skip_pad_rows(desc)) ? ids : nullptr, (gather && skip_pad_rows(desc)) ? bq_hm : nullptr,
                              nullptr, nullptr, nullptr, s, &pdrop_attn, desc->precision != NRMS_PRECISION_FP32);
    if (rc) return rc;
    if (wo) {
        NTArgs o{};
        o.M = M; o.N = d; o.K = d; o.rows_per_tile = NT_BM;
        o.A = acts->attn; o.lda = d; o.W = w->w_o; o.bias = w->b_o; o.C = acts->ctx; o.ldc = d; o.drop = drop_c;
        rc = nt_gemm(desc, A_PLAIN, E_STORE, o, wplanes, s, "out_proj_fwd");
        if (rc) return rc;
    }
    if (wide_additive(desc)) {
        // Z = ctx Wa^T + ba by the plain NT GEMM into the T buffer, then tanh / scores / softmax / pooling per sequence
        NRMS_REQUIRE(acts->t != nullptr, "encoder_fwd: acts.t is required for q_dim > 256 or d_model > 512 (also for inference)");
        NTArgs z{};
        z.M = M; z.N = q; z.K = d; z.rows_per_tile = NT_BM;
        z.A = acts->ctx; z.lda = d; z.W = w->w_add; z.bias = w->b_add; z.C = acts->t; z.ldc = q;
        rc = nt_gemm(desc, A_PLAIN, E_STORE, z, wplanes, s, "addattn_proj_fwd");
        if (rc) return rc;
        return launch_addattn_rows_fwd_wide(desc->n_seq, S, d, q, acts->t, w->q_vec, acts->ctx, pmask, acts->w, out, s);
    }
    const int npass = desc->precision == NRMS_PRECISION_FP32 ? 0 : (desc->precision == NRMS_PRECISION_BF16X3 ? 3 : 1);
    return launch_addattn_fwd(desc->n_seq, S, d, q, acts->ctx, w->w_add, w->b_add, w->q_vec, acts->t, acts->w, out,
                              pmask, npass, wplanes, s);
}

extern "C" size_t nrms_encoder_fwd_scratch_bytes(const nrms_encoder_desc* desc) {
    if (validate_desc(desc, "encoder_fwd_scratch_bytes")) return 0;
    if (desc->precision == NRMS_PRECISION_FP16) return fwd16_scratch(desc).total;
    return fwd_scratch(desc).total;
}

extern "C" size_t nrms_encoder_fused_qkv_bytes(const nrms_encoder_desc* desc) {
    if (desc == nullptr || !(desc->flags & NRMS_FLAG_FUSED_SEQ64) || validate_desc(desc, "encoder_fused_qkv_bytes")) return 0;
    return user64_qkv_bytes(desc->n_seq, desc->n_heads);
}

extern "C" size_t nrms_encoder_bwd_workspace_bytes(const nrms_encoder_desc* desc) {
    if (validate_desc(desc, "encoder_bwd_workspace_bytes")) return 0;
    if (desc->precision == NRMS_PRECISION_FP16) return bwd16_layout(desc).total;
    return bwd_layout(desc).total;
}

// ---- helper streams (declared in common.h) -------------------------------------------------------------------------
namespace nrms {
struct SideEntry { int device; hipStream_t caller; SideSet* set; };
static std::mutex g_side_mu;
static std::vector<SideEntry> g_side_sets;

SideSet* side_streams_for(hipStream_t caller) {
    if (getenv("NRMS_NO_SIDE_STREAMS") != nullptr) return nullptr;        // read per call: a profiler pass serialises the step
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_side_mu);
    for (const SideEntry& e : g_side_sets)
        if (e.device == dev && e.caller == caller) return e.set;
    if (g_side_sets.size() >= 32) return nullptr;
    SideSet* ss = new SideSet();
    bool ok = true;
    for (int i = 0; i < 2 && ok; ++i) ok = hipStreamCreateWithFlags(&ss->s[i], hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 8 && ok; ++i) ok = hipEventCreateWithFlags(&ss->ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {                                     // remember the failure (a null set): do not retry on every call
        for (int i = 0; i < 2; ++i) if (ss->s[i]) (void)hipStreamDestroy(ss->s[i]);
        for (int i = 0; i < 8; ++i) if (ss->ev[i]) (void)hipEventDestroy(ss->ev[i]);
        delete ss;
        ss = nullptr;
        (void)hipGetLastError();
    }
    g_side_sets.push_back(SideEntry{dev, caller, ss});
    return ss;
}

int side_order(SideSet* ss, int e, hipStream_t from, hipStream_t to, const char* what) {
    if (hipEventRecord(ss->ev[e], from) != hipSuccess || hipStreamWaitEvent(to, ss->ev[e], 0) != hipSuccess) {
        set_error("%s: ordering the helper stream failed: %s", what, hipGetErrorString(hipGetLastError()));
        return NRMS_ELAUNCH;
    }
    return NRMS_OK;
}

SideJoinGuard::~SideJoinGuard() {
    if (!armed) return;
    for (int i = 0; i < 2; ++i)                    // best effort on an error path: events 6 and 7 are reserved for this
        if (hipEventRecord(ss->ev[6 + i], ss->s[i]) == hipSuccess) (void)hipStreamWaitEvent(caller, ss->ev[6 + i], 0);
}
}  // namespace nrms

// step 5 of the backward (shared by nrms_encoder_bwd and nrms_encoder_bwd_wqkv)
// One helper stream for the weight-gradient (TN) GEMMs of the fp32 / split-bf16 backward: d(W_add), d(W_O) and d(W_qkv)
// feed nothing inside the call, so they run beside the main stream's chain (d(ctx) GEMM, attention backward -- a long
// latency-bound kernel that leaves most of the matrix pipe idle --, dX GEMM, scatter).  In order among themselves (they
// share the partial-slab workspace), forked from the main stream where their inputs are complete, joined at the end of
// the call (on every exit path: SideJoinGuard).  The streams belong to the caller's stream (side_streams_for): backwards
// on different streams or devices of one process do not share them.
static int bwd_wqkv(const nrms_encoder_desc* desc, const float* xin, const float* dqkv, const int* n_live,
                    const nrms_encoder_grads* grads, float* tn_partial, hipStream_t s) {
    const int d = desc->d_model, M = desc->n_seq * desc->seq_len;
    TNArgs t{};
    t.M = M; t.N = 3 * d; t.K = d; t.amode = A_PLAIN;
    t.A = dqkv; t.lda = 3 * d; t.B = xin; t.ldb = d;
    t.dW = grads->w_qkv; t.dbias = grads->b_qkv; t.partial = tn_partial;
    t.perm = HeadPerm{d / desc->n_heads, desc->n_heads};          // dQKV columns are head-major
    if (skip_pad_rows(desc)) t.m_dev = n_live;                    // compact operands
    return tn_gemm(desc, t, s, "dwqkv_bwd");
}

extern "C" int nrms_encoder_bwd_wqkv(const nrms_encoder_desc* desc, const int64_t* ids, const float* x,
                                     const nrms_encoder_acts* acts, const nrms_encoder_grads* grads, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    int rc = validate_desc(desc, "encoder_bwd_wqkv");
    if (rc) return rc;
    if (desc->precision == NRMS_PRECISION_FP16) return fused_bwd16_join((hipStream_t)stream);   // the GEMMs run already: order them
    NRMS_REQUIRE(acts && grads && workspace, "encoder_bwd_wqkv: null argument");
    NRMS_REQUIRE(grads->w_qkv && grads->b_qkv, "encoder_bwd_wqkv: null gradient buffer");
    const bool gather = desc->vocab > 0;
    NRMS_REQUIRE(gather ? (ids != nullptr && acts->x != nullptr) : (x != nullptr), "encoder_bwd_wqkv: input missing");
    if (desc->n_seq == 0) return NRMS_OK;
    const BwdWorkspace L = bwd_layout(desc);
    NRMS_REQUIRE(workspace_bytes >= L.total, "encoder_bwd_wqkv: workspace %zu < required %zu bytes", workspace_bytes, L.total);
    char* base = (char*)workspace;
    return bwd_wqkv(desc, gather ? acts->x : x, (const float*)(base + L.dqkv), (const int*)(base + L.n_live), grads,
                    (float*)(base + L.tn_partial), (hipStream_t)stream);
}

extern "C" int nrms_encoder_bwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int64_t* ids,
                                const float* x, const uint8_t* mask, const nrms_encoder_acts* acts, const float* dout,
                                const nrms_encoder_grads* grads, float* dx, void* workspace, size_t workspace_bytes,
                                void* stream) {
    int rc = validate_desc(desc, "encoder_bwd");
    if (rc) return rc;
    if (desc->precision == NRMS_PRECISION_FP16)
        return encoder_bwd16(desc, w, ids, x, acts, dout, grads, dx, workspace, workspace_bytes, (hipStream_t)stream);
    NRMS_REQUIRE(w && acts && dout && grads && workspace, "encoder_bwd: null argument");
    NRMS_REQUIRE(acts->qkv && acts->ctx && acts->t && acts->w, "encoder_bwd: all saved activations are required");
    NRMS_REQUIRE(grads->w_qkv && grads->b_qkv && grads->w_add && grads->b_add && grads->q_vec,
                 "encoder_bwd: null gradient buffer");
    const bool gather = desc->vocab > 0, wo = desc->use_output_proj != 0;
    NRMS_REQUIRE(gather ? (ids != nullptr && acts->x != nullptr && grads->table != nullptr) : (x != nullptr && dx != nullptr),
                 "encoder_bwd: %s missing", gather ? "ids/acts.x/grads.table" : "x/dx");
    NRMS_REQUIRE(!wo || (w->w_o && acts->attn && grads->w_o && grads->b_o),
                 "encoder_bwd: use_output_proj needs w_o, acts.attn, grads.w_o, grads.b_o");
    NRMS_REQUIRE(desc->mask_mode == 0 || mask != nullptr, "encoder_bwd: mask_mode=%d but mask is null", desc->mask_mode);
    const BwdWorkspace L = bwd_layout(desc);
    if (workspace_bytes < L.total) {
        set_error("encoder_bwd: workspace %zu < required %zu bytes", workspace_bytes, L.total);
        return NRMS_EWORKSPACE;
    }
    NRMS_REQUIRE(((uintptr_t)workspace & 255) == 0, "encoder_bwd: workspace must be 256-byte aligned");
    if (desc->n_seq == 0) return NRMS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int S = desc->seq_len, d = desc->d_model, q = desc->q_dim, M = desc->n_seq * S;
    const Dropout drop_e = make_dropout(desc->seed, desc->p_drop_embed);
    const Dropout drop_c = make_dropout(desc->seed, desc->p_drop_ctx);
    const Dropout no_drop = make_dropout(0, 0.f);
    const uint8_t* amask = (desc->mask_mode & 1) ? mask : nullptr;
    const uint8_t* pmask = (desc->mask_mode & 2) ? mask : nullptr;
    char* base = (char*)workspace;
    float* dctx = (float*)(base + L.dctx);
    float* dqkv = (float*)(base + L.dqkv);
    float* dattn = (float*)(base + L.dattn);
    float* ds = (float*)(base + L.ds);
    float* wqkv_t = (float*)(base + L.wqkv_t);
    float* wadd_t = (float*)(base + L.wadd_t);
    float* wo_t = (float*)(base + L.wo_t);
    float* tn_partial = (float*)(base + L.tn_partial);
    float* dq_partial = (float*)(base + L.dq_partial);
    void* wplanes = (void*)(base + L.wplanes);

    // a deferred fp16 join still pending on this stream (NRMS_FLAG_DEFER_WQKV without nrms_encoder_bwd_wqkv yet): order it
    // first -- the events and helper streams below are the same set
    rc = fused_bwd16_join(s);
    if (rc) return rc;
    SideSet* ss = side_streams_for(s);
    const bool side = ss != nullptr;
    hipStream_t s2 = side ? ss->s[0] : s;
    SideJoinGuard join_guard(ss, s);
    auto fork = [&](int i) -> int {               // the helper stream continues from here on the main stream
        return side ? side_order(ss, i, s, s2, "encoder_bwd") : NRMS_OK;
    };

    const bool fused64 = (desc->flags & NRMS_FLAG_FUSED_SEQ64) != 0;
    if (fused64) {
        // steps 1, 2 and 4 as ONE kernel (csrc/user64.hip): pooling backward, d(ctx), attention backward -> ds, d(q_vec), dQKV
        rc = launch_user64_bwd(desc->n_seq, S, d, desc->n_heads, q, w->w_add, w->q_vec, base + L.u64planes, dout, acts->t, acts->w,
                               acts->qkv, ds, dq_partial, grads->q_vec, dqkv, s);
        if (rc) return rc;
    } else {
        // 1. pooling rows: ds, d(q_vec)
        if (wide_additive(desc))
            rc = launch_addattn_rows_bwd_wide(desc->n_seq, S, d, q, acts->ctx, dout, acts->w, acts->t, ds, dq_partial, grads->q_vec,
                                              pmask, s);
        else
            rc = launch_addattn_bwd_rows(desc->n_seq, S, d, q, acts->ctx, dout, acts->w, acts->t, ds, dq_partial, grads->q_vec,
                                         pmask, s);
        if (rc) return rc;
    }
    // 3. d(w_add), d(b_add) = dZ^T [ctx | 1]: needs ds only -- on the helper stream, beside everything below
    rc = fork(0);
    if (rc) return rc;
    {
        TNArgs t{};
        t.M = M; t.N = q; t.K = d; t.amode = A_DZ;
        t.ds = ds; t.qv = w->q_vec; t.T = acts->t; t.B = acts->ctx; t.ldb = d;
        t.dW = grads->w_add; t.dbias = grads->b_add; t.partial = tn_partial;
        rc = tn_gemm(desc, t, s2, "dwadd_bwd");
        if (rc) return rc;
    }
    if (!fused64) {
        // 2. d(ctx) = dZ Wa + w_s dout, then through the context-dropout mask in the GEMM's coalesced
        //    epilogue (one Philox call per float4) -- the attention backward then carries no RNG work.
        rc = launch_transpose(w->w_add, wadd_t, q, d, s);
        if (rc) return rc;
        {
            NTArgs g{};
            g.M = M; g.N = d; g.K = q; g.rows_per_tile = NT_BM;
            g.ds = ds; g.qv = w->q_vec; g.T = acts->t;
            g.W = wadd_t; g.C = dctx; g.ldc = d;
            g.wrow = acts->w; g.dout = dout; g.S = S;
            g.drop = drop_c;
            rc = nt_gemm(desc, A_DZ, E_DCTX, g, wplanes, s, "dctx_bwd");
            if (rc) return rc;
        }
}
    const float* dattn_in = dctx;
    if (wo) {
        // 3b. output projection: d(W_O), d(b_O) = dC^T [attn | 1] (helper stream: reads d(ctx));  d(attn) = dC W_O
        rc = fork(1);
        if (rc) return rc;
        TNArgs t{};
        t.M = M; t.N = d; t.K = d; t.amode = A_PLAIN;
        t.A = dctx; t.lda = d; t.B = acts->attn; t.ldb = d;
        t.dW = grads->w_o; t.dbias = grads->b_o; t.partial = tn_partial;
        rc = tn_gemm(desc, t, s2, "dwo_bwd");
        if (rc) return rc;
        if (side && hipEventRecord(ss->ev[3], s2) != hipSuccess) {      // d(ctx) may be overwritten after this (step 6)
            set_error("encoder_bwd: hipEventRecord failed");
            return NRMS_ELAUNCH;
        }
        rc = launch_transpose(w->w_o, wo_t, d, d, s);
        if (rc) return rc;
        NTArgs g{};
        g.M = M; g.N = d; g.K = d; g.rows_per_tile = NT_BM;
        g.A = dctx; g.lda = d; g.W = wo_t; g.C = dattn; g.ldc = d;
        rc = nt_gemm(desc, A_PLAIN, E_STORE, g, wplanes, s, "dattn_bwd");
        if (rc) return rc;
        dattn_in = dattn;
    }
    // Token compaction (news encoder).  dX is consumed only by the embedding scatter, which skips padding
    // tokens (padding_idx = 0, nrms_v0.py:134-136) -- in a MIND-shaped batch about two thirds of the
    // token rows -- so the dX GEMM and the scatter run on the non-padding rows only.  With
    // NRMS_FLAG_PAD_ROW_ZERO the same holds for d(w_qkv): acts.x is compact already and the attention
    // backward writes dQKV compact, handing the padding rows' share of d(b_qkv) over as column sums.
    int* live = (int*)(base + L.live);
    int* pos = (int*)(base + L.pos);
    int* n_live = (int*)(base + L.n_live);
    const bool compact = skip_pad_rows(desc);
    const HeadPerm perm{d / desc->n_heads, desc->n_heads};
    if (gather) {
        rc = launch_compact_live_rows((long)M, ids, live, compact ? pos : nullptr, n_live, (int*)(base + L.cscr), s);
        if (rc) return rc;
    }
    // 4. attention backward
    Dropout pdrop_attn = make_dropout(desc->seed, desc->p_drop_attn);
    pdrop_attn.seq_index = desc->seq_index; pdrop_attn.seq_h = desc->n_heads;
    if (fused64)
        rc = NRMS_OK;                               // (dQKV is there already)
    else if (wide_attention(desc))
        rc = launch_attention_wide(true, desc->n_seq, S, d, desc->n_heads, acts->qkv, nullptr, pdrop_attn, dattn_in, dqkv, amask, s);
    else
        rc = launch_attention(true, desc->n_seq, S, d, desc->n_heads, acts->qkv, nullptr, no_drop, dattn_in, dqkv, amask,
                              nullptr, nullptr, compact ? pos : nullptr, compact ? (float*)(base + L.padsum) : nullptr,
                              compact ? grads->b_qkv : nullptr, s, &pdrop_attn, desc->precision != NRMS_PRECISION_FP32);
    if (rc) return rc;
    // 5. d(w_qkv), d(b_qkv) = dQKV^T [X | 1]   (X = the forward's gathered+dropped embeddings) -- or later,
    //    by nrms_encoder_bwd_wqkv (NRMS_FLAG_DEFER_WQKV)
    if ((desc->flags & NRMS_FLAG_DEFER_WQKV) == 0) {
        rc = fork(2);                              // after the attention backward AND its padding-row sums into d(b_qkv)
        if (rc) return rc;
        rc = bwd_wqkv(desc, gather ? acts->x : x, dqkv, n_live, grads, tn_partial, s2);
        if (rc) return rc;
    }
    // 6. dX = dQKV Wqkv.  User encoder: that is the answer.  News encoder: compact dX into the (now
    //    dead) dctx buffer, then the compact scatter.
    rc = launch_transpose(w->w_qkv, wqkv_t, 3 * d, d, s, perm);
    if (rc) return rc;
    {
        NTArgs g{};
        g.M = M; g.N = d; g.K = 3 * d; g.rows_per_tile = NT_BM;
        g.A = dqkv; g.lda = 3 * d; g.W = wqkv_t;
        g.C = gather ? dctx : dx; g.ldc = d;
        if (gather) { g.a_rows = compact ? nullptr : live; g.m_dev = n_live; }
        if (side && wo && gather && hipStreamWaitEvent(s, ss->ev[3], 0) != hipSuccess) {   // dX lands in the d(ctx) buffer d(W_O) reads
            set_error("encoder_bwd: hipStreamWaitEvent failed");
            return NRMS_ELAUNCH;
        }
        rc = nt_gemm(desc, A_PLAIN, E_STORE, g, wplanes, s, "dx_bwd");
        if (rc) return rc;
    }
    if (gather) {
        const bool atomic_scatter = getenv("NRMS_ATOMIC_SCATTER") != nullptr;      // A/B switch, read per call (no latched state)
        if (atomic_scatter) rc = launch_scatter_dropout_compact((long)M, d, ids, live, n_live, dctx, drop_e, grads->table, s);
        else rc = launch_scatter_grouped((long)M, desc->vocab, d, ids, live, n_live, dctx, drop_e, grads->table,
                                         (int*)(base + L.sscr), s);
    }
    if (side && rc == NRMS_OK) {                   // the caller's stream continues after the weight gradients too
        rc = side_order(ss, 4, s2, s, "encoder_bwd");
        if (rc == NRMS_OK) join_guard.disarm();
    }
    return rc;
}

extern "C" int nrms_layernorm_fwd(int64_t n_rows, int32_t d, const float* x, const float* gamma, const float* beta, float eps,
                                  float* y, float* stats, void* stream) {
    NRMS_REQUIRE(n_rows >= 0 && d > 0 && eps > 0.f, "layernorm_fwd: n_rows=%ld d=%d eps=%g", (long)n_rows, d, (double)eps);
    NRMS_REQUIRE(n_rows == 0 || (x && gamma && beta && y), "layernorm_fwd: null argument");
    return launch_layernorm_fwd((long)n_rows, d, x, gamma, beta, eps, y, stats, (hipStream_t)stream);
}

extern "C" size_t nrms_layernorm_bwd_workspace_bytes(int32_t d) { return d > 0 ? layernorm_bwd_workspace_floats(d) * sizeof(float) : 0; }

extern "C" int nrms_layernorm_bwd(int64_t n_rows, int32_t d, const float* x, const float* gamma, const float* stats,
                                  const float* dy, float* dx, float* dgamma_dbeta, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    NRMS_REQUIRE(n_rows >= 0 && d > 0, "layernorm_bwd: n_rows=%ld d=%d", (long)n_rows, d);
    NRMS_REQUIRE(n_rows == 0 || (x && gamma && stats && dy && dx && dgamma_dbeta && workspace), "layernorm_bwd: null argument");
    if (workspace_bytes < nrms_layernorm_bwd_workspace_bytes(d)) {
        set_error("layernorm_bwd: workspace %zu < required %zu bytes", workspace_bytes, nrms_layernorm_bwd_workspace_bytes(d));
        return NRMS_EWORKSPACE;
    }
    return launch_layernorm_bwd((long)n_rows, d, x, gamma, stats, dy, dx, dgamma_dbeta, (float*)workspace, (hipStream_t)stream);
}

static int features_args(const nrms_news_features* f, FeatArgs* a, const char* who) {
    NRMS_REQUIRE(f != nullptr, "%s: null descriptor", who);
    NRMS_REQUIRE(f->n >= 0 && f->d_text > 0 && f->d_cat > 0 && f->n_cat > 0 && f->n_sub > 0, "%s: bad sizes", who);
    NRMS_REQUIRE(f->p_drop >= 0.f && f->p_drop < 1.f, "%s: p_drop must be in [0,1)", who);
    NRMS_REQUIRE((long)f->n * (2 * f->d_text + 2 * f->d_cat) < (1L << 40), "%s: too many elements", who);
    NRMS_REQUIRE(f->n == 0 || (f->categ && f->subcateg), "%s: null ids", who);
    a->n = (long)f->n; a->dt = f->d_text; a->dc = f->d_cat; a->n_cat = f->n_cat; a->n_sub = f->n_sub;
    a->title = f->title_vec; a->abst = f->abst_vec; a->cat_table = f->cat_table; a->sub_table = f->sub_table;
    a->categ = f->categ; a->subcateg = f->subcateg;
    a->drop = make_dropout(f->seed, f->p_drop);
    return NRMS_OK;
}

extern "C" int nrms_news_features_fwd(const nrms_news_features* f, float* out, void* stream) {
    FeatArgs a{};
    int rc = features_args(f, &a, "news_features_fwd");
    if (rc) return rc;
    NRMS_REQUIRE(f->n == 0 || (f->title_vec && f->abst_vec && f->cat_table && f->sub_table && out), "news_features_fwd: null argument");
    a.out = out;
    return launch_features_fwd(a, (hipStream_t)stream);
}

extern "C" int nrms_news_features_bwd(const nrms_news_features* f, const float* dout, float* d_title_vec, float* d_abst_vec,
                                      float* d_cat_table, float* d_sub_table, void* stream) {
    FeatArgs a{};
    int rc = features_args(f, &a, "news_features_bwd");
    if (rc) return rc;
    NRMS_REQUIRE(f->n == 0 || (dout && d_title_vec && d_abst_vec && d_cat_table && d_sub_table), "news_features_bwd: null argument");
    a.dout = dout; a.d_title = d_title_vec; a.d_abst = d_abst_vec; a.d_cat_table = d_cat_table; a.d_sub_table = d_sub_table;
    return launch_features_bwd(a, (hipStream_t)stream);
}

extern "C" int nrms_sanitize_ids(const int64_t* src, int64_t* dst, int64_t n, int32_t vocab, int32_t* n_bad, void* stream) {
    NRMS_REQUIRE(n >= 0 && vocab > 0, "sanitize_ids: n=%ld vocab=%d", (long)n, vocab);
    NRMS_REQUIRE(n == 0 || (src && dst && n_bad), "sanitize_ids: null argument");
    return launch_sanitize_ids((long)n, src, false, dst, vocab, n_bad, (hipStream_t)stream);
}

extern "C" int nrms_sanitize_ids_i32(const int32_t* src, int64_t* dst, int64_t n, int32_t vocab, int32_t* n_bad, void* stream) {
    NRMS_REQUIRE(n >= 0 && vocab > 0, "sanitize_ids_i32: n=%ld vocab=%d", (long)n, vocab);
    NRMS_REQUIRE(n == 0 || (src && dst && n_bad), "sanitize_ids_i32: null argument");
    return launch_sanitize_ids((long)n, src, true, dst, vocab, n_bad, (hipStream_t)stream);
}

extern "C" size_t nrms_sequence_partition_count_ints(int32_t n_seq) { return n_seq >= 0 ? title_order_cnt_ints(n_seq) : 0; }

extern "C" int nrms_sequence_partition(const int64_t* ids, int32_t n_seq, int32_t seq_len, int32_t* order, int32_t* counts, void* stream) {
    NRMS_REQUIRE(n_seq >= 0 && seq_len >= 1 && seq_len <= 64, "sequence_partition: n_seq=%d seq_len=%d", n_seq, seq_len);
    NRMS_REQUIRE(counts != nullptr, "sequence_partition: null counts");
    if (n_seq == 0) return hipMemsetAsync(counts, 0, 2 * sizeof(int), (hipStream_t)stream) == hipSuccess ? NRMS_OK : NRMS_ELAUNCH;
    NRMS_REQUIRE(ids && order, "sequence_partition: null argument");
    return launch_title_order(n_seq, seq_len, ids, order, counts, (hipStream_t)stream, 2);
}

extern "C" int nrms_title_dedup(const int64_t* ids, int64_t n_titles, int32_t seq_len, int32_t* table, int64_t table_size,
                                int32_t* inverse, int32_t* rep_rows, int32_t* n_unique, void* stream) {
    NRMS_REQUIRE(n_titles >= 0 && n_titles < (1L << 30) && seq_len >= 1, "title_dedup: n_titles=%ld seq_len=%d", (long)n_titles, seq_len);
    NRMS_REQUIRE(n_unique != nullptr, "title_dedup: null n_unique");
    NRMS_REQUIRE(n_titles == 0 || (ids && table && inverse && rep_rows), "title_dedup: null argument");
    NRMS_REQUIRE(n_titles == 0 || (table_size >= 2 * n_titles && (table_size & (table_size - 1)) == 0),
                 "title_dedup: table_size=%ld must be a power of two >= 2 * n_titles (%ld)", (long)table_size, (long)n_titles);
    return launch_title_dedup((long)n_titles, seq_len, ids, table, (long)table_size, inverse, rep_rows, n_unique, (hipStream_t)stream);
}

extern "C" void nrms_timing_enable(int enable) {
    std::lock_guard<std::mutex> lk(g_tmu);
    g_timing = enable != 0;
}

extern "C" void nrms_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_tmu);
    for (auto& t : g_launches) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    g_launches.clear();
}

extern "C" int nrms_timing_read(const char* prefix, double* total_ms, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_tmu);
    double tot = 0.0;
    int64_t n = 0;
    const std::string p = prefix ? prefix : "";
    for (auto& t : g_launches) {
        if (t.name.compare(0, p.size(), p) != 0) continue;
        if (hipEventSynchronize(t.stop) != hipSuccess) { set_error("timing_read: event sync failed"); return NRMS_ELAUNCH; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.start, t.stop) != hipSuccess) { set_error("timing_read: elapsed failed"); return NRMS_ELAUNCH; }
        tot += ms;
        ++n;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return NRMS_OK;
}

extern "C" const char* nrms_last_error(void) { return g_err; }
extern "C" const char* nrms_version(void) { return "nrms_hip 0.3 (gfx950, fp32 MFMA; nrms_v0 + nrms_v1 topologies)"; }
