/*
 * nrms_hip.h -- C ABI of libnrms_hip.so: the MI355X (gfx950) NRMS train/eval hot path.
 *
 * This is the drop-in boundary beneath the reference's Python plugin API
 * (model/__init__.py:22-23 `import_module('model.'+name).Model(config)`, called as
 * `outputs = model(datas)` at train_eval.py:111,242,323).  The reference has no native
 * layer; every entry point below names the reference lines whose ATen op sequence it
 * replaces (paths relative to /root/reference/MIND_2020/).
 *
 * Conventions
 *   - plain C: POD structs, raw DEVICE pointers, sizes; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the default stream).
 *   - the library allocates nothing and owns nothing: activations, gradients and workspace
 *     are caller buffers; workspace sizes come from the *_workspace_bytes queries.
 *   - every call is asynchronous on `stream`, re-entrant across streams, and returns
 *     0 on success or a negative NRMS_E* code; nrms_last_error() gives the (thread-local) text.
 *     Helper streams: nrms_encoder_bwd forks its weight-gradient GEMMs (and the fp16 calls their id-only / weight-only
 *     bookkeeping) onto two helper streams that belong to the pair (device, caller `stream`): created the first time that
 *     stream makes such a call, never shared between caller streams, forked from and joined back into `stream` with events
 *     inside the call -- before it returns, on the error paths too, or in nrms_encoder_bwd_wqkv under NRMS_FLAG_DEFER_WQKV.
 *     Several backwards may therefore be in flight in one process as long as they are on DIFFERENT streams (two engines on
 *     two streams train concurrently); calls on the SAME stream must be issued by one host thread at a time, as stream order
 *     demands anyway.  A pending deferred join (NRMS_FLAG_DEFER_WQKV, fp16) is flushed by the next nrms_encoder_fwd /
 *     nrms_encoder_bwd on that stream in ANY precision, so a caller that never calls nrms_encoder_bwd_wqkv still gets
 *     correct ordering.  The environment variable NRMS_NO_SIDE_STREAMS keeps everything on `stream`.
 *   - all matrices are row-major and dense; fp32 unless stated.  M = n_seq * seq_len.
 *   - gradients are ACCUMULATED (+=) into the caller's buffers (zero them per step, as
 *     `model.zero_grad()` does at train_eval.py:115).
 */
#ifndef NRMS_HIP_H
#define NRMS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRMS_OK            0
#define NRMS_EINVAL       -1   /* bad dims / null pointer / unsupported shape */
#define NRMS_ELAUNCH      -2   /* HIP launch or runtime error */
#define NRMS_EWORKSPACE   -3   /* workspace too small */

#define NRMS_PRECISION_FP32   0 /* f32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32): exact fp32 */
#define NRMS_PRECISION_BF16X3 1 /* dense projections as split-bf16: x = hi + lo, hi*hi + hi*lo + lo*hi on
                                   v_mfma_f32_16x16x32_bf16, fp32 accumulate (~2^-16 relative per product);
                                   attention, softmaxes, pooling, loss, optimizer and all HBM tensors stay fp32 */
#define NRMS_PRECISION_BF16   2 /* same kernels, hi*hi only: plain bf16 inputs, fp32 accumulate */
#define NRMS_FP16_KP 320        /* fp16 mode: pitch of x (input features, zero padded) */
#define NRMS_FP16_DP 320        /* pitch of ctx (32 per head, zero padded) */
#define NRMS_FP16_QP 224        /* pitch of t */
#define NRMS_PRECISION_FP16   3 /* fused path: one wavefront per sequence, every contraction on v_mfma_f32_32x32x16_f16
                                   (fp16 operands: 11 significant bits, fp32 accumulate), Q/K/V, attention probabilities
                                   and tanh(.) register-resident, activations kept for the backward in fp16.
                                   Restrictions: seq_len <= 64, d_model <= 316, d_k <= 32, n_heads <= 10, q_dim <= 224,
                                   no output projection, no masks (NRMS_EINVAL otherwise).  With use_output_proj (the
                                   news encoder of nrms_v1, model/nrms_v1.py:109-162: heads wider than 32 + W_O,
                                   dropout after W_O): vocab > 0 with NRMS_FLAG_PAD_ROW_ZERO and no embedding dropout,
                                   seq_len <= 32, 32 < d_k <= 50, 3 n_heads + 1 <= 19, n_heads (d_k - 48) <= 16,
                                   d_model <= 316 a multiple of 10 (d_model / 10 <= 32), q_dim <= 224; the context
                                   dropout then acts on the ten [d_model / 10]-wide blocks of the projection's output
                                   (same padded [M, 320] counter layout).  Activation buffers change
                                   meaning (see nrms_encoder_acts); context-dropout counters run over the padded
                                   [M, NRMS_FP16_DP] layout, 32 columns per head, in the 16-bit-field scheme
                                   (nrms_dropout_keep_mask with d = 320 and site 1 | NRMS_DROPOUT_FIELDS16). */

/* nrms_encoder_desc.flags */
/* The caller guarantees that row 0 of `table` (the padding row, nn.Embedding padding_idx=0,
 * nrms_v0.py:134-136) is all zeros.  A padding token's embedding is then exactly zero (also under
 * dropout), so its Q|K|V row is exactly the bias and it adds nothing to d(w_qkv): the projection and its
 * weight-gradient GEMM run on the non-padding tokens only.  Results are identical to the dense path.
 * The row keeps its value under training (its gradient is identically zero), so the flag is a property
 * of the loaded weights.  Ignored when vocab == 0. */
#define NRMS_FLAG_PAD_ROW_ZERO 1
/* nrms_encoder_bwd leaves d(w_qkv) / d(b_qkv) out (its step 5); the caller runs nrms_encoder_bwd_wqkv with the
 * same descriptor, activations and workspace afterwards.  Everything else, in particular the embedding-table
 * gradient, is complete when nrms_encoder_bwd returns: a data-parallel caller starts the all-reduce of the
 * table gradient (95 % of the gradient bytes) there and lets it run underneath the deferred GEMM.
 * NRMS_PRECISION_FP16: the weight-gradient GEMMs (d(w_qkv), d(b_qkv), d(w_add), d(b_add)) are already running on the
 * library's helper streams when nrms_encoder_bwd returns; with this flag the call does not wait for them (dx and the table
 * gradient are complete in stream order, those four are NOT) and nrms_encoder_bwd_wqkv orders them into the stream. */
#define NRMS_FLAG_DEFER_WQKV 2
/* nrms_encoder_bwd, NRMS_PRECISION_FP16 news encoder: `acts.scratch` is still exactly what the nrms_encoder_fwd call of this
 * step left in it (the caller has not passed that buffer to another forward in between).  The backward then reads the token
 * and title lists the forward built there (live rows, row positions, the three title classes) instead of rebuilding them
 * from the ids (five small launches).  Without the flag nothing in acts.scratch is read by the backward.  The promise is
 * checked on the host: the library remembers which forward (scratch, ids, n_seq, seq_len, ...) last built lists in a scratch
 * buffer and forgets it when ANY nrms_encoder_fwd is handed that buffer again; a backward whose arguments do not match the
 * record rebuilds the lists as if the flag were absent (correct results, five more launches). */
#define NRMS_FLAG_FWD_SCRATCH_KEPT 4
/* User encoder (vocab == 0), NRMS_PRECISION_BF16X3, 33 <= seq_len <= 64, d_model <= 300, even d_k <= 32, n_heads <= 10,
 * q_dim <= 224, no output projection, no mask, no dropout: the whole pass runs as ONE kernel per direction (csrc/user64.hip: a
 * wave per 32-row half of a history, Q / K / V / probabilities / tanh in registers, every product in split-bf16) instead of the
 * projection GEMM -> attention -> additive-attention chain; the weight-gradient and dX GEMMs of the backward are unchanged.
 * Under the flag acts.qkv is NOT [M, 3d] floats but nrms_encoder_fused_qkv_bytes(desc) bytes of operand fragments (internal
 * layout, written by the forward, read by the backward); acts.ctx, acts.t, acts.w keep their meaning (and may all be NULL in
 * inference).  Set on both nrms_encoder_fwd and nrms_encoder_bwd of a pass, or on neither; NRMS_EINVAL for any other shape. */
#define NRMS_FLAG_FUSED_SEQ64 8

/* One self-attention + additive-pooling encoder pass over n_seq sequences of seq_len rows.
 * vocab > 0  : news encoder -- input is `ids` [n_seq, seq_len] int64, rows gathered from
 *              `table` (NewsEncoder.forward, model/nrms_v0.py:154-176).
 * vocab == 0 : user encoder -- input is `x` [n_seq, seq_len, d_model]
 *              (UserEncoder.forward, model/nrms_v0.py:188-199). */
typedef struct nrms_encoder_desc {
    int32_t  n_seq;        /* titles (B*(H+C)) or users (B) */
    int32_t  seq_len;      /* L or H, 1..64 */
    int32_t  d_model;      /* config.word_embed_size (nrms_naml user encoder: news_feature_size); multiple of 4, <= 1024 */
    int32_t  n_heads;      /* config.num_attention_heads (v1 / naml news encoder: title_heads_num; naml user encoder:
                              user_heads_num); d_k = d_model / n_heads <= 128.  Even d_k <= 64 runs on the MFMA attention
                              kernels, anything else on the shape-general fp32 kernel (csrc/wide.hip) */
    int32_t  q_dim;        /* config.query_vector_dim (naml user encoder: query_vector_dim_large), multiple of 4, <= 512;
                              above 256 (or d_model > 512) acts.t is required in inference too */
    int32_t  vocab;        /* rows of `table`, or 0 */
    float    p_drop_embed; /* dropout on the gathered embeddings (nrms_v0.py:137); 0 in eval / nrms_v1 */
    float    p_drop_ctx;   /* dropout on the attention output (nrms_v0.py:171-173; nrms_v1.py:161 after W_O) */
    int32_t  precision;    /* NRMS_PRECISION_* */
    int32_t  use_output_proj; /* 1: nrms_v1 topology, MHSA ends in output_linear W_O (nrms_v1.py:55,80) */
    int32_t  mask_mode;    /* bit 0: pairwise attention mask mask_i*mask_j -> -1e9 (nrms_v1.py:27-33);
                              bit 1: additive-attention mask -> -1e9 (nrms_v1.py:100-101); 0 = nrms_v0 */
    int32_t  flags;        /* NRMS_FLAG_*; 0 = none */
    uint64_t seed;         /* counter-based RNG key for the dropout masks (per step) */
    float    loss_scale;   /* NRMS_PRECISION_FP16 backward only: the fp16 gradient tensors are carried multiplied by a
                              power of two and the results divided by it.  <= 0 (recommended): chosen on the device per
                              call as the power of two that puts max |dout| into [64, 128), so no loss reduction, batch
                              size or world size can overflow or flush the fp16 tensors -- 2^9 of head room for the tensors
                              derived from dout (dZ, d(ctx), dQKV, dX).  Weights of unusual norm can use that up: the
                              gradients then come out inf / nan, nrms_adam_step_guarded / nrms_grad_guard count them, and
                              the caller backs off with loss_scale = -n (a negative integer, n <= 24): n more powers of
                              two of head room, max |dout| into [64, 128) / 2^n.  > 0: used as given */
    float    p_drop_attn;  /* dropout on the attention PROBABILITIES (nrms_naml.py:36-39, dropout site 2; 0 in nrms_v0 /
                              nrms_v1); not combinable with NRMS_PRECISION_FP16.  With NRMS_FLAG_PAD_ROW_ZERO an
                              all-padding sequence keeps a closed form: context of query i = b_v x (kept keys of i) /
                              (seq_len (1 - p)) */
    const int32_t* seq_index; /* optional (device) [n_seq]: the call holds a COMPACTED batch and sequence r is sequence
                              seq_index[r] of the full one -- the attention-probability dropout then draws the full batch's
                              decisions (counters by seq_index[r], not r).  fp32 / bf16x3 / bf16 chain only, no embedding or
                              context dropout; null = 0 .. n_seq - 1.  See nrms_sequence_partition / nrms_encoder_empty_fwd */
} nrms_encoder_desc;

/* Parameters, in the reference's own tensor layout ([out,in] Linear weights).
 * w_qkv is W_Q.weight, W_K.weight, W_V.weight stacked on dim 0 (nrms_v0.py:35-37). */
typedef struct nrms_encoder_weights {
    const float* table;    /* [vocab, d]   news_encoder.word_embedding.0.weight, or NULL */
    const float* w_qkv;    /* [3d, d] */
    const float* b_qkv;    /* [3d] */
    const float* w_o;      /* [d, d]       output_linear.weight (nrms_v1.py:55) or NULL */
    const float* b_o;      /* [d] */
    const float* w_add;    /* [q, d]       additive_attention.linear.weight (nrms_v0.py:91) */
    const float* b_add;    /* [q] */
    const float* q_vec;    /* [q]          additive_attention.attention_query_vector (:92-93) */
} nrms_encoder_weights;

typedef struct nrms_encoder_grads {   /* same shapes as the weights; accumulated */
    float* table;          /* dense [vocab, d]; row 0 never written (padding_idx=0, nrms_v0.py:136) */
    float* w_qkv;
    float* b_qkv;
    float* w_o;            /* NULL unless use_output_proj */
    float* b_o;
    float* w_add;
    float* b_add;
    float* q_vec;
} nrms_encoder_grads;

/* Activations the forward saves for the backward (caller-owned).  In eval only `qkv`,
 * `ctx` (and `x` for the news encoder) are required scratch; `t` and `w` may then be NULL. */
typedef struct nrms_encoder_acts {
    float* x;              /* [M, d]   news encoder only: gathered word embeddings AFTER dropout
                                       (the user encoder's input is the caller's `x`); may be NULL if vocab==0.
                                       With NRMS_FLAG_PAD_ROW_ZERO only the non-padding tokens are stored, compact,
                                       in ascending token order (the buffer is still sized [M, d]) */
    float* qkv;            /* [M, 3d]  Q|K|V projections incl. bias, columns HEAD-MAJOR: head h occupies columns
                                       [3 d_k h, 3 d_k (h+1)) as Q | K | V (internal layout, read by the backward
                                       only).  With NRMS_FLAG_PAD_ROW_ZERO the rows of sequences that consist of
                                       padding only are left unwritten (the attention takes their closed form) */
    float* attn;           /* [M, d]   use_output_proj only: head-concatenated attention output (input of W_O) */
    float* ctx;            /* [M, d]   head-concatenated attention output AFTER dropout */
    float* t;              /* [M, q]   tanh(linear(ctx))            (nrms_v0.py:108) */
    float* w;              /* [M]      additive-attention softmax weights (nrms_v0.py:110-112) */
    void*  scratch;        /* nrms_encoder_fwd_scratch_bytes(desc) bytes: head-major W_qkv copy, bf16 weight planes,
                              token compaction lists */
    /* NRMS_PRECISION_FP16: x, ctx, t hold fp16 with the FIXED pitches KP = 320, DP = 320, QP = 224
     * (NRMS_FP16_KP / _DP / _QP):  x [M + 1, KP] (required for both encoders: gathered embeddings / the cast input),
     * ctx [Mp, DP] and t [Mp, QP] with Mp = n_seq * (seq_len <= 32 ? 32 : 64) rows (every sequence padded to whole
     * 32-row blocks, internal fragment order); w [M] fp32; qkv is unused (may be NULL); attn is unused unless
     * use_output_proj, then fp16 [Mp, DP] (the head concatenation in the operand order of the W_O tiles; read by the
     * backward's d(w_o) product). */
} nrms_encoder_acts;

/* Forward: embedding gather(+dropout) -> QKV projection -> per-head softmax(QK^T/sqrt(d_k))V
 * [-> output projection W_O] -> dropout -> tanh(linear)·q -> softmax over the sequence -> weighted sum.
 * Replaces model/nrms_v0.py:13-23,46-76,100-126,154-176,188-199 and, with use_output_proj / mask_mode,
 * model/nrms_v1.py:15-105,128-162,208-211.
 * out: [n_seq, d].  Exactly one of ids / x is used (by desc->vocab). */
size_t nrms_encoder_fwd_scratch_bytes(const nrms_encoder_desc* desc);
/* Size of acts.qkv under NRMS_FLAG_FUSED_SEQ64 (0 if the descriptor is not eligible for it). */
size_t nrms_encoder_fused_qkv_bytes(const nrms_encoder_desc* desc);
int nrms_encoder_fwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w,
                     const int64_t* ids, const float* x, const uint8_t* mask /* [n_seq, seq_len] or NULL */,
                     const nrms_encoder_acts* acts, float* out, void* stream);

/* Backward of the above (autograd through the same lines; `loss.backward()` train_eval.py:126).
 * dout: [n_seq, d].  grads: accumulated.  dx: [M, d] gradient w.r.t. `x` (user encoder), or
 * NULL for the news encoder, whose input gradient is scatter-added into grads->table
 * (the dense embedding gradient the reference builds 55x per step). */
size_t nrms_encoder_bwd_workspace_bytes(const nrms_encoder_desc* desc);
int nrms_encoder_bwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w,
                     const int64_t* ids, const float* x, const uint8_t* mask,
                     const nrms_encoder_acts* acts, const float* dout,
                     const nrms_encoder_grads* grads, float* dx,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Step 5 of the backward on its own: d(w_qkv), d(b_qkv) += dQKV^T [X | 1] from the dQKV left in `workspace`
 * by nrms_encoder_bwd(desc with NRMS_FLAG_DEFER_WQKV) -- same desc, ids / x, acts, grads, workspace. */
int nrms_encoder_bwd_wqkv(const nrms_encoder_desc* desc, const int64_t* ids, const float* x,
                          const nrms_encoder_acts* acts, const nrms_encoder_grads* grads,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Word ids must lie in [0, vocab) before they reach nrms_encoder_fwd / _bwd: the kernels index the table, the
 * token histogram and the placement lists with the raw id.  nn.Embedding raises on an out-of-range index
 * (nrms_v0.py:134-139,166); this is the device-side counterpart for untrusted input:
 * dst[i] = src[i] if 0 <= src[i] < vocab, else 0 (the padding id); *n_bad += number of ids replaced
 * (device int32 the caller zeroes and reads back when it chooses to synchronise).  dst may alias src. */
int nrms_sanitize_ids(const int64_t* src, int64_t* dst, int64_t n, int32_t vocab, int32_t* n_bad, void* stream);
/* The same from int32 ids (a feed that keeps its ids in 32 bits moves half the bytes; the encoder entry points take the
 * validated int64 copy either way). */
int nrms_sanitize_ids_i32(const int32_t* src, int64_t* dst, int64_t n, int32_t vocab, int32_t* n_bad, void* stream);

/* Evaluation encodes every DISTINCT title once (get_news_vector, nrms_v0.py:278-289, is the reference's hook for
 * caching news vectors; an impression padded to max_candidate_size = 300 slots, data_handler.py:174-177, is mostly
 * padding and repeats).  Exact grouping of the n_titles rows of ids [n_titles, seq_len] (seq_len = 1 groups plain news
 * ids): inverse[t] = index in [0, *n_unique) of the group of row t, rep_rows[u] = one row of group u (groups are
 * numbered in arrival order, which may differ between runs; the groups themselves do not).  table: caller-provided
 * workspace of table_size int32, a power of two >= 2 * n_titles.  *n_unique is a device int32. */
int nrms_title_dedup(const int64_t* ids, int64_t n_titles, int32_t seq_len, int32_t* table, int64_t table_size,
                     int32_t* inverse, int32_t* rep_rows, int32_t* n_unique, void* stream);

/* Click scores: bmm(cand [B,C,d], user [B,d,1]) then masked_fill(mask==0, -1e9)
 * (DotProductClickPredictor, nrms_v0.py:205-216; mask nrms_v0.py:272-274).  mask may be NULL. */
int nrms_click_score_fwd(int32_t B, int32_t C, int32_t d, const float* cand, const float* user,
                         const uint8_t* mask, float* scores, void* stream);
/* The same scores with the candidate vectors named by row index into a table of news vectors instead of being
 * materialised: scores[b][c] = <news_vec[index[b*C + c]], user[b]> (the evaluation path: 300 candidate slots per
 * impression over a few thousand distinct news).  index: int32 [B*C], every entry in [0, n_vec). */
int nrms_click_score_indexed(int32_t B, int32_t C, int32_t d, const float* news_vec, int64_t n_vec, const int32_t* index,
                             const float* user, const uint8_t* mask, float* scores, void* stream);
/* dscores [B,C] -> dcand [B,C,d], duser [B,d] (masked slots receive no gradient). */
int nrms_click_score_bwd(int32_t B, int32_t C, int32_t d, const float* cand, const float* user,
                         const uint8_t* mask, const float* dscores, float* dcand, float* duser,
                         void* stream);

/* nn.CrossEntropyLoss with every label 0 (train_eval.py:63,116-117):
 * loss_sum[0] += sum_b -log_softmax(scores[b])[0]   (caller divides by the global batch),
 * dscores = (softmax(scores) - onehot0) * grad_scale (grad_scale = 1/B_global).
 * dscores may be NULL (loss only). */
int nrms_ce_loss_fwd_bwd(int32_t B, int32_t C, const float* scores, float* loss_sum,
                         float* dscores, float grad_scale, void* stream);

/* torch.optim.Adam defaults (train_eval.py:48,127): one fused pass over a flat fp32 buffer.
 * g is multiplied by grad_scale first (1/world_size after a summing all-reduce).
 * step is 1-based.  lr / betas / eps are doubles, as torch holds them (1 - beta is rounded to fp32 once). */
int nrms_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   double lr, double beta1, double beta2, double eps, int32_t step, float grad_scale,
                   void* stream);

/* The same update, but an element whose (scaled) gradient is inf or nan is left out -- param, exp_avg and exp_avg_sq keep
 * their values there, so one overflowing fp16 backward cannot poison Adam's moments for the rest of training -- and counted:
 * *n_nonfinite (device int32, caller-zeroed) += number of elements skipped.  The caller reads the counter when it chooses to
 * (asynchronously) and lowers the fp16 loss scale (nrms_encoder_desc.loss_scale = -n).  Data parallel: every rank applies
 * this to the same reduced gradient and skips the same elements. */
int nrms_adam_step_guarded(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                           double lr, double beta1, double beta2, double eps, int32_t step, float grad_scale,
                           int32_t* n_nonfinite, void* stream);
/* For callers that run their own optimizer (torch.optim.Adam on the autograd path, train_eval.py:126-127): inf / nan elements
 * of grad [n] are replaced by 0 in place and counted into *n_nonfinite (device int32, caller-zeroed). */
int nrms_grad_guard(size_t n, float* grad, int32_t* n_nonfinite, void* stream);

/* Per-impression AUC of evaluate() (train_eval.py:219-227,255-271 + evaluation.py:26-27):
 * scores, labels [n_imp, max_c] (padded), lens [n_imp] = candidates actually shown;
 * auc[i] = roc_auc_score(labels[i,:lens[i]], scores[i,:lens[i]]) in float64 (NaN if one class only). */
int nrms_impression_auc(int32_t n_imp, int32_t max_c, const float* scores, const uint8_t* labels,
                        const int32_t* lens, double* auc, void* stream);

/* ---- nrms_naml pieces around the two encoder passes (model/nrms_naml.py) ----
 * LayerNorm over the last dimension (nn.LayerNorm(news_feature_size) on the history vectors, nrms_naml.py:207,238):
 * y = (x - mean) / sqrt(var + eps) * gamma + beta, biased variance.  stats [n_rows, 2] = (mean, 1/std) is written when
 * non-NULL (the backward needs it). */
int nrms_layernorm_fwd(int64_t n_rows, int32_t d, const float* x, const float* gamma, const float* beta, float eps,
                       float* y, float* stats, void* stream);
size_t nrms_layernorm_bwd_workspace_bytes(int32_t d);
/* dx [n_rows, d] is overwritten; d(gamma) [d] and d(beta) [d] are ACCUMULATED into dgamma_dbeta [2d] (gamma's gradient
 * first: norm.weight and norm.bias are adjacent in the caller's flat gradient buffer).  Reproducible (fixed-order sums). */
int nrms_layernorm_bwd(int64_t n_rows, int32_t d, const float* x, const float* gamma, const float* stats, const float* dy,
                       float* dx, float* dgamma_dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* News feature rows (NewsEncoder.forward, nrms_naml.py:168-175):
 * out[n] = dropout([title_vec[n] | abst_vec[n] | cat_table[categ[n]] | sub_table[subcateg[n]]]), dropout site 3 over
 * [n, 2 d_text + 2 d_cat].  categ / subcateg: int64 [n], already validated (nrms_sanitize_ids with the table's row count). */
typedef struct nrms_news_features {
    int64_t n;               /* slots: B*(H+C) */
    int32_t d_text;          /* config.word_embed_size */
    int32_t d_cat;           /* config.cate_embed_size */
    int32_t n_cat, n_sub;    /* config.category_nums, config.subcategory_nums (table rows; row 0 = padding_idx) */
    float   p_drop;          /* config.dropout in training, 0 in eval */
    uint64_t seed;
    const float* title_vec;  /* [n, d_text] */
    const float* abst_vec;   /* [n, d_text] */
    const float* cat_table;  /* [n_cat, d_cat] */
    const float* sub_table;  /* [n_sub, d_cat] */
    const int64_t* categ;    /* [n] */
    const int64_t* subcateg; /* [n] */
} nrms_news_features;
int nrms_news_features_fwd(const nrms_news_features* f, float* out, void* stream);
/* dout [n, 2 d_text + 2 d_cat] -> d_title_vec, d_abst_vec [n, d_text] (overwritten; d_title_vec also holds the table sums' per-chunk
 * partial results during the call); the table gradients are ACCUMULATED into d_cat_table / d_sub_table, row 0 (padding_idx = 0,
 * nrms_naml.py:107-108) untouched; atomic-free (slot chunks in ascending order). */
int nrms_news_features_bwd(const nrms_news_features* f, const float* dout, float* d_title_vec, float* d_abst_vec,
                           float* d_cat_table, float* d_sub_table, void* stream);

/* ---- Gather + additive-attention aggregate over variable-length segments (BASELINE configs 4-5) ----
 *   out[s] = sum_{k in [seg_ptr[s], seg_ptr[s+1])} alpha_k x[idx[k]],   alpha = softmax over the segment of
 *   q_vec . tanh(W_add x[idx[k]] + b_add)          -- AdditiveAttention (model/nrms_v0.py:100-126) over an index list.
 * It is the aggregation step of a HieRec-style hierarchical interest model (clicked news pooled per sub-topic, sub-topic interests
 * per topic, topic interests per user: each level is this call over a PARTITION of its rows) and of a user-news graph encoder
 * (neighbour gather + attention aggregate: segments = adjacency lists, a row may be a member of many segments).  The reference
 * holds no implementation of either (model/tanr.py is empty): the checker is a torch restatement of the formula
 * (oracle/segpool_oracle.py) -- PARITY UNPINNED.  An empty segment yields a zero row. */
#define NRMS_SEGPOOL_ROWS_UNIQUE 1   /* every row is a member of at most one segment: d(logit) and dx are then plain stores by the
                                        segment's wave; without the flag the backward sorts the list entries by row (stable) and
                                        adds a row's shares in ascending list position -- bit-reproducible either way */
typedef struct nrms_segpool_desc {
    int64_t n_rows;       /* rows of x */
    int64_t n_seg;        /* segments */
    int64_t nnz;          /* members in all segments = seg_ptr[n_seg] */
    int32_t d;            /* row width: multiple of 4, <= 1024 */
    int32_t q;            /* width of the additive attention: multiple of 4, <= 512 */
    int32_t precision;    /* NRMS_PRECISION_FP32 / _BF16X3 / _BF16: the arithmetic of the two projections (x W_add^T and dZ W_add, dZ^T x) */
    int32_t flags;        /* NRMS_SEGPOOL_* */
} nrms_segpool_desc;
size_t nrms_segment_pool_workspace_bytes(const nrms_segpool_desc* desc);
/* x [n_rows, d]; seg_ptr int32 [n_seg + 1] ascending from 0; idx int32 [nnz], every entry in [0, n_rows).  Saved for the backward:
 * t [n_rows, q] = tanh(x W_add^T + b_add), alpha [nnz]; logit [n_rows] is scratch the caller provides.  out [n_seg, d]. */
int nrms_segment_pool_fwd(const nrms_segpool_desc* desc, const float* x, const float* w_add, const float* b_add,
                          const float* q_vec, const int32_t* seg_ptr, const int32_t* idx, float* t, float* logit, float* alpha,
                          float* out, void* workspace, size_t workspace_bytes, void* stream);
/* dout [n_seg, d] -> dx [n_rows, d] (OVERWRITTEN: rows outside every segment get their projection gradient, i.e. zero);
 * dw_add [q, d], db_add [q], dq_vec [q] are ACCUMULATED. */
int nrms_segment_pool_bwd(const nrms_segpool_desc* desc, const float* x, const float* w_add, const float* q_vec,
                          const int32_t* seg_ptr, const int32_t* idx, const float* t, const float* alpha, const float* dout,
                          float* dx, float* dw_add, float* db_add, float* dq_vec, void* workspace, size_t workspace_bytes,
                          void* stream);

/* Index lists from padded neighbour lists (a sampled sub-graph as [n_seg, K] int64 with -1 for "no neighbour"): entries inside
 * [0, n_rows) are kept in their order.  seg_ptr [n_seg + 1], idx [capacity n_seg * K]; pass nnz = n_seg * K (a capacity) in the
 * nrms_segpool_desc -- the lists' real length is seg_ptr[n_seg], on the device.  Without NRMS_SEGPOOL_ROWS_UNIQUE (a row listed
 * by several segments, as in a graph) the backward sorts the list entries by row (stable) and adds a row's contributions in
 * ascending list position: no atomics, the same bits on every run. */
int nrms_csr_from_padded(int64_t n_seg, int32_t K, const int64_t* lists, int64_t n_rows, int32_t* seg_ptr, int32_t* idx, void* stream);

/* ---- All-padding sequences of the output-projection topology in closed form (csrc/empty_seq.hip; nrms_naml's word-level
 * encoder, model/nrms_naml.py:42-100,121-177: 41 % of a MIND-shaped batch's title / abstract slots are history padding).  With a
 * zero padding row every Q | K | V row of such a sequence is the bias, so attention row i is b_v scaled per head by
 * c_ih = (kept keys of query i) / (seq_len (1 - p_drop_attn)), and W_O, the additive attention and their gradients collapse to
 * products with n_heads + 1 vectors that depend on the weights only.  The caller splits its sequences with nrms_sequence_partition
 * (order[0 .. counts[0]) = sequences with a real token, order[n_seq .. n_seq + counts[1]) = all-padding ones; counts holds
 * nrms_sequence_partition_count_ints(n_seq) ints, device), runs nrms_encoder_fwd / _bwd on the first list's sequences (gathered,
 * desc.seq_index = that list) and these two calls on the second: desc.n_seq = counts[1], seq_index = the second list, out / dout
 * [desc.n_seq, d_model] compact.  Same desc rules as the chain (use_output_proj = 1, no mask, no embedding / context dropout;
 * seq_len <= 64, n_heads <= 8, d_model <= 512, q_dim <= 256); gradients are ACCUMULATED into grads (w_add, b_add, q_vec, w_o, b_o,
 * and the V third of b_qkv; every other gradient of such a sequence is exactly zero), per-wave partial sums in a fixed order. */
size_t nrms_sequence_partition_count_ints(int32_t n_seq);
int nrms_sequence_partition(const int64_t* ids, int32_t n_seq, int32_t seq_len, int32_t* order /* [2 * n_seq] */, int32_t* counts, void* stream);
size_t nrms_encoder_empty_workspace_bytes(const nrms_encoder_desc* desc);
size_t nrms_encoder_empty_saved_bytes(const nrms_encoder_desc* desc);
/* saved: nrms_encoder_empty_saved_bytes(desc) bytes that the forward fills (the rows' kept-key factors and pooling weights) and the
 * backward of the same sequences reads; null in the forward = inference. */
int nrms_encoder_empty_fwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int32_t* seq_index, float* out,
                           void* saved, void* workspace, size_t workspace_bytes, void* stream);
int nrms_encoder_empty_bwd(const nrms_encoder_desc* desc, const nrms_encoder_weights* w, const int32_t* seq_index, const float* dout,
                           const void* saved, const nrms_encoder_grads* grads, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Index side of a HieRec-style hierarchical interest model (BASELINE configs[3]; SURVEY f-4; PARITY UNPINNED: no reference
 * implementation, checked against oracle/segpool_oracle.py).  A user's clicked news (H <= 64 history slots, `valid` = the
 * batch dict's browsed_mask) are grouped by sub-topic id, the sub-topic groups by topic id, in order of first occurrence; every
 * level's groups live in fixed per-user slots b * H + g.  nrms_hier_tree_build writes the three index lists
 * nrms_segment_pool_fwd / _bwd aggregate over:
 *   level 1: rows = history slots b * H + k;       l1_ptr [B * H + 1], l1_idx [B * H]: one segment per sub-topic group slot
 *   level 2: rows = sub-topic group slots;         l2_ptr [B * H + 1], l2_idx [B * H]: one segment per topic group slot
 *   level 3: rows = topic group slots;             l3_ptr [B + 1],     l3_idx [B * H]: one segment per user
 * and per slot the ids and click counts (l1_sub, l1_top, l1_cnt, l2_top, l2_cnt: int32 [B * H]; count 0 = empty slot),
 * n_valid [B].  Every level partitions its rows (NRMS_SEGPOOL_ROWS_UNIQUE). */
size_t nrms_hier_tree_scratch_bytes(int32_t B, int32_t H);
int nrms_hier_tree_build(int32_t B, int32_t H, const uint8_t* valid, const int64_t* topic, const int64_t* subtopic,
                         int32_t* l1_ptr, int32_t* l1_idx, int32_t* l1_sub, int32_t* l1_top, int32_t* l1_cnt,
                         int32_t* l2_ptr, int32_t* l2_idx, int32_t* l2_top, int32_t* l2_cnt, int32_t* l3_ptr,
                         int32_t* l3_idx, int32_t* n_valid, void* scratch, size_t scratch_bytes, void* stream);
/* interest of an occupied slot = its aggregate + the embedding of its (sub-)topic: u[slot] += table[id[slot]] where cnt[slot] > 0
 * (table [n_ids, d]; an id outside it adds nothing and takes no gradient);
 * backward: dtable[r] += sum of du over the occupied slots with id == r, in a fixed order (per-chunk partial sums in `workspace`,
 * then the chunks in ascending order; no atomics). */
int nrms_hier_add_embedding_fwd(int64_t n_slots, int32_t d, int32_t n_ids, const int32_t* id, const int32_t* cnt, const float* table,
                                float* u, void* stream);
size_t nrms_hier_add_embedding_bwd_workspace_bytes(int64_t n_slots, int32_t d, int32_t n_ids);
int nrms_hier_add_embedding_bwd(int64_t n_slots, int32_t d, int32_t n_ids, const int32_t* id, const int32_t* cnt,
                                const float* du, float* dtable, void* workspace, size_t workspace_bytes, void* stream);
/* Hierarchical matching.  nrms_hier_match: for candidate (b, c) the user's sub-topic / topic group slot with the candidate's ids
 * (-1: the user never clicked there) and the share of the user's clicks in it.  nrms_hier_score_fwd:
 *   score = l_s f_s <n, u1[sub_slot]> + l_t f_t <n, u2[top_slot]> + (1 - l_s - l_t) <n, ug[b]>,  masked slots -1e9.
 * _bwd: dcand, dug overwritten; du1, du2 [B * H, d] ACCUMULATED (zero them first); one wavefront per user, no atomics. */
int nrms_hier_match(int32_t B, int32_t C, int32_t H, const int64_t* cand_topic, const int64_t* cand_subtopic,
                    const int32_t* l1_sub, const int32_t* l1_cnt, const int32_t* l2_top, const int32_t* l2_cnt,
                    const int32_t* n_valid, int32_t* sub_slot, float* sub_frac, int32_t* top_slot, float* top_frac, void* stream);
int nrms_hier_score_fwd(int32_t B, int32_t C, int32_t d, const float* cand, const float* u1, const float* u2, const float* ug,
                        const int32_t* sub_slot, const float* sub_frac, const int32_t* top_slot, const float* top_frac,
                        const uint8_t* mask, float lambda_sub, float lambda_top, float* scores, void* stream);
int nrms_hier_score_bwd(int32_t B, int32_t C, int32_t d, const float* cand, const float* u1, const float* u2, const float* ug,
                        const int32_t* sub_slot, const float* sub_frac, const int32_t* top_slot, const float* top_frac,
                        const uint8_t* mask, float lambda_sub, float lambda_top, const float* dscores, float* dcand, float* du1,
                        float* du2, float* dug, void* stream);

/* The keep mask (1 = kept) the encoder kernels apply at a dropout site, for n_rows x d
 * elements: site 0 = embedding dropout (nrms_v0.py:137), site 1 = context dropout (:171-173), site 2 = attention
 * probabilities ([n_seq * n_heads * seq_len, seq_len], nrms_naml.py:36-39), site 3 = news feature rows (:175).
 * Lets a test replay a training step through the oracle with identical masks. */
#define NRMS_DROPOUT_FIELDS16 0x100   /* or-ed into `site`: the 16-bit-field scheme of the fp16 mode's context dropout --
                                        one Philox call per 8 elements of the flat [n_rows, d] layout, P(drop) = p rounded
                                        down to a multiple of 2^-16.  The fp16 kernels apply it over the padded [tokens, 320]
                                        layout with the two middle bits of a column's index inside its 32-column head block
                                        swapped: mask of column 16a + 8b + 4c + e = field at 16a + 8c + 4b + e */
int nrms_dropout_keep_mask(uint64_t seed, int32_t site, int64_t n_rows, int32_t d, float p_drop,
                           uint8_t* keep, void* stream);

/* Per-kernel device timing (HIP events on the launch stream), for bench.py's roofline leg.
 * nrms_timing_read synchronises the recorded events; returns 0 and the accumulated
 * milliseconds / launch count of kernels whose name starts with `prefix`. */
void nrms_timing_enable(int enable);
void nrms_timing_reset(void);
int  nrms_timing_read(const char* prefix, double* total_ms, int64_t* launches);

const char* nrms_last_error(void);
const char* nrms_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NRMS_HIP_H */
