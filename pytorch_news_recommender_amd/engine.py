"""Host-side driver of the HIP hot path: owns the device buffers (via torch, which is only
plumbing here: allocator + streams) and issues the C-ABI calls of include/nrms_hip.h in the
order of one NRMS step.

Layout in HBM (all fp32, row-major):
  flat parameter buffer  [ table V*d | news: Wqkv 3d*d, bqkv 3d, Wa q*d, ba q, qv q | user: same ]
  flat gradient buffer   same layout (one RCCL all-reduce / one fused Adam pass over it)
  titles                 ids [N, L] int64, N = B*H history titles (user-major) then B*C candidates
  news vectors           [N, d]: rows [0, B*H) are the user-encoder input, rows [B*H, N) the candidates
  saved activations      qkv [M,3d], ctx [M,d], t [M,q], w [M] per encoder (M = sequences * length)
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import torch

from . import _lib
from .synth import ENCODERS


@dataclass(frozen=True)
class ModelDims:
    n_words: int
    word_embed_size: int
    num_attention_heads: int            # user-encoder heads (and news-encoder heads unless news_heads is set)
    query_vector_dim: int
    news_heads: int = 0                 # nrms_v1: config.title_heads_num for the news encoder
    output_proj: bool = False           # nrms_v1 topology: MHSA ends in output_linear (W_O)
    style: str = "v0"                   # parameter naming: "v0" (model/nrms_v0.py) or "v1" (model/nrms_v1.py)

    def heads(self, enc):
        return self.news_heads if (enc == "news_encoder" and self.news_heads) else self.num_attention_heads


def role_names(style: str, output_proj: bool):
    """Ordered (parameter name, encoder, role) triples; the order IS the flat-buffer order."""
    out = []
    if style == "v0":
        out.append(("news_encoder.word_embedding.0.weight", "news_encoder", "table"))
    else:
        out.append(("news_encoder.word_embedding.weight", "news_encoder", "table"))
    for enc in ENCODERS:
        if style == "v0":
            a = enc + ".multihead_self_attention."
            qkv = [a + "W_Q", a + "W_K", a + "W_V"]
            wo = a + "W_O"
            qv = enc + ".additive_attention.attention_query_vector"
        else:
            a = enc + ".multi_head_self_attention."
            qkv = [a + "linear_layers.%d" % i for i in range(3)]
            wo = a + "output_linear"
            qv = enc + ".additive_attention.query_vector"
        out += [(n + ".weight", enc, r) for n, r in zip(qkv, ("wq", "wk", "wv"))]
        out += [(n + ".bias", enc, r) for n, r in zip(qkv, ("bq", "bk", "bv"))]
        if output_proj:
            out += [(wo + ".weight", enc, "wo"), (wo + ".bias", enc, "bo")]
        out += [(enc + ".additive_attention.linear.weight", enc, "wa"),
                (enc + ".additive_attention.linear.bias", enc, "ba"), (qv, enc, "qv")]
    return out


class FlatLayout:
    """Offsets (in floats) of the reference-named tensors inside the flat buffer.
    W_Q/W_K/W_V (and their biases) are adjacent so the kernels see one [3d, d] matrix."""

    def __init__(self, dims: ModelDims):
        V, d, q = dims.n_words, dims.word_embed_size, dims.query_vector_dim
        if d % 4 or q % 4:
            raise ValueError("word_embed_size and query_vector_dim must be multiples of 4")
        self.dims = dims
        self.entries = {}
        self.blocks = {enc: {} for enc in ENCODERS}
        shapes = {"table": (V, d), "wq": (d, d), "wk": (d, d), "wv": (d, d), "bq": (d,), "bk": (d,), "bv": (d,),
                  "wo": (d, d), "bo": (d,), "wa": (q, d), "ba": (q,), "qv": (q,)}
        off = 0
        for name, enc, role in role_names(dims.style, dims.output_proj):
            shp = shapes[role]
            n = 1
            for x in shp:
                n *= x
            self.entries[name] = (off, shp, n)
            self.blocks[enc][role] = off
            off += n
        self.total = off
        self.names = list(self.entries)
        for enc in ENCODERS:
            b = self.blocks[enc]
            assert b["wk"] == b["wq"] + d * d and b["wv"] == b["wq"] + 2 * d * d      # adjacency the kernels rely on
            assert b["bk"] == b["bq"] + d and b["bv"] == b["bq"] + 2 * d
        self.table = self.blocks["news_encoder"]["table"]

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shp, n = self.entries[name]
        return flat[off:off + n].view(shp)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class NRMSEngine:
    """One NRMS forward / backward / optimizer step on one GPU."""

    def __init__(self, dims: ModelDims, device, precision="fp32"):
        self.lib = _lib.load()
        self.dims = dims
        self.set_precision(precision)
        self.layout = FlatLayout(dims)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NrmsError("the NRMS HIP engine needs a GPU device (got %s); there is no CPU path" % device)
        self._bufs = {}
        self._saved = None
        self.fp16_user_encoder = False     # precision "fp16": the user encoder runs in bf16x3 unless this is set
        self.fp16_backward = True          # training in fp16 mode runs the fused fp16 backward (csrc/fused16_bwd.hip)
        self.loss_scale = 0.0              # fp16 backward: 0 = chosen on the device from max |dout| per call (nrms_hip.h)
        # precision "fp16", nrms_v1's W_O + wide-head news encoder on csrc/fused16_v1.hip: OPT-IN (config.fp16_v1_news_encoder).
        # Its scores sit up to 1.56e-4 from fp32 on a 512-user batch (2 % of them beyond north_star's absolute 1e-4),
        # so the default keeps v1's news encoder in bf16x3 (1e-6)
        self.fp16_wide_heads = False
        # precision "fp16": passes that keep nothing for a backward (evaluate / test, get_news_vector, forward under no_grad) run
        # in bf16x3 unless this is set (config.fp16_inference): evaluation scores then sit ~1e-6 from the reference and the
        # per-impression AUC cannot move by rank flips of near-tied candidates (round 3 measured 1.4e-4 on 1 024 impressions)
        self.fp16_inference = False
        # the user encoder's bf16x3 passes (33..64-slot histories, d <= 300, heads <= 32 wide, no mask / W_O) as ONE kernel per
        # direction (csrc/user64.hip, NRMS_FLAG_FUSED_SEQ64) instead of the GEMM -> attention -> additive chain; False = the chain
        self.fused_user_encoder = True
        self.fp16_wide_heads_backward = True    # ... its training step too (csrc/fused16_v1_bwd.hip); False: training in bf16x3
        self._gen = 0                      # generation stamp of _saved (checked by the autograd backward)
        # out-of-range word ids: counted on the device by nrms_sanitize_ids, surfaced without a host sync
        # (the count is copied to pinned memory behind the kernel and looked at on a later call)
        self._bad_ids = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._bad_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._bad_event = None
        self._news_cache = None
        # fp16 backward: non-finite gradient elements (an overflow of the loss-scaled fp16 tensors) are skipped and counted on
        # the device by the guarded optimizer / grad_guard; the count reaches the host asynchronously (like the id check) and
        # lowers the loss scale for the following steps: desc.loss_scale = -loss_scale_backoff (include/nrms_hip.h)
        self._grad_bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._grad_bad_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._grad_bad_event = None
        self.loss_scale_backoff = 0        # extra powers of two of head room below the default [64, 128) target
        self.grad_overflow_steps = 0       # steps whose gradient held non-finite elements (seen so far)
        self._clean_polls = 0

    FP16_LIMITS = dict(seq_len=64, d_model=316, d_k=32, n_heads=10, q_dim=224)

    def _fp16_ok(self, enc, seq_len, mask_mode, training):
        """The fused fp16 kernels cover the shapes of include/nrms_hip.h (NRMS_PRECISION_FP16); an encoder pass
        outside them runs in bf16x3 instead.  So does the user encoder unless fp16_user_encoder is set: it is 3 % of
        the flops, and with it in split-bf16 (~2^-16 relative) the scores keep a margin inside north_star's 1e-4
        and the user encoder's additive-attention gradients (cancelling sums) are no longer fp16 noise."""
        d, L = self.dims, self.FP16_LIMITS
        h = d.heads(enc)
        if enc == "user_encoder" and not self.fp16_user_encoder:
            return False
        if not training and not self.fp16_inference:
            return False
        if d.word_embed_size % h:
            return False
        if d.output_proj:
            # nrms_v1's news encoder (heads of 33..50 columns + W_O): csrc/fused16_v1.hip, padding-skipping path only
            dk = d.word_embed_size // h
            return (enc == "news_encoder" and self.fp16_wide_heads and self.pad_row_zero and not mask_mode
                    and seq_len <= 32 and d.word_embed_size <= L["d_model"] and d.word_embed_size % 10 == 0
                    and d.word_embed_size // 10 <= 32 and 32 < dk <= 50 and 3 * h + 1 <= 19 and h * max(dk - 48, 0) <= 16
                    and d.query_vector_dim <= L["q_dim"] and (not training or self.fp16_wide_heads_backward))
        return (seq_len <= L["seq_len"] and d.word_embed_size <= L["d_model"] and h <= L["n_heads"]
                and d.word_embed_size // h <= L["d_k"] and d.query_vector_dim <= L["q_dim"]
                and not d.output_proj and not mask_mode and (not training or self.fp16_backward))

    def set_precision(self, precision):
        """"fp32" (exact f32 MFMA), "bf16x3" (split-bf16 projections, ~2^-16 relative), "bf16", or "fp16"
        (fused one-wave-per-sequence kernels on fp16 MFMA, fp16 activations -- v0's news encoder and, with output_proj, nrms_v1's
        six-heads-of-50 + W_O news encoder; bf16x3 where a shape, a mask or the user encoder is outside them)."""
        if precision not in _lib.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        self.precision = precision
        # NRMS_FLAG_PAD_ROW_ZERO: set by the owner of the parameters when embedding row 0 is all zeros
        # (include/nrms_hip.h); False = dense path
        self.pad_row_zero = False

    # ---- buffers ----------------------------------------------------------------------
    def _buf(self, key, numel, dtype=torch.float32):
        t = self._bufs.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(max(int(numel), 1), dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def _desc(self, enc, n_seq, seq_len, p_embed=0.0, p_ctx=0.0, seed=0, mask_mode=0, training=False):
        d = self.dims
        prec = self.precision
        if prec == "fp16" and not self._fp16_ok(enc, seq_len, mask_mode, training):
            prec = "bf16x3"
        flags = _lib.NRMS_FLAG_PAD_ROW_ZERO if (self.pad_row_zero and enc == "news_encoder") else 0
        h = d.heads(enc)
        if (enc == "user_encoder" and self.fused_user_encoder and prec == "bf16x3" and not d.output_proj and not mask_mode
                and 32 < seq_len <= 64 and d.word_embed_size <= 300 and d.word_embed_size % h == 0
                and (d.word_embed_size // h) <= 32 and (d.word_embed_size // h) % 2 == 0 and h <= 10 and d.query_vector_dim <= 224):
            flags |= _lib.NRMS_FLAG_FUSED_SEQ64
        return _lib.EncoderDesc(n_seq=n_seq, seq_len=seq_len, d_model=d.word_embed_size, n_heads=h,
                                q_dim=d.query_vector_dim, vocab=d.n_words if enc == "news_encoder" else 0,
                                p_drop_embed=float(p_embed), p_drop_ctx=float(p_ctx),
                                precision=_lib.PRECISIONS[prec], use_output_proj=int(d.output_proj),
                                mask_mode=int(mask_mode),
                                flags=flags,
                                seed=int(seed), loss_scale=float(self.loss_scale), p_drop_attn=0.0)

    def _ptrs(self, cls, flat, enc):
        b = self.layout.blocks[enc]
        base = flat.data_ptr()
        p = lambda role: (base + 4 * b[role]) if role in b else None
        return cls(table=p("table"), w_qkv=p("wq"), b_qkv=p("bq"), w_o=p("wo"), b_o=p("bo"), w_add=p("wa"),
                   b_add=p("ba"), q_vec=p("qv"))

    def _weights(self, flat, enc):
        return self._ptrs(_lib.EncoderWeights, flat, enc)

    def _grads(self, gflat, enc):
        return self._ptrs(_lib.EncoderGrads, gflat, enc)

    def _acts(self, tag, M, need_bwd, gather=False, desc=None):
        d, q = self.dims.word_embed_size, self.dims.query_vector_dim
        if desc is not None and desc.precision == _lib.NRMS_PRECISION_FP16:
            # fp16 activations with padded pitches (include/nrms_hip.h, nrms_encoder_acts)
            KP, DP, QP = _lib.NRMS_FP16_KP, _lib.NRMS_FP16_DP, _lib.NRMS_FP16_QP
            h = torch.float16
            x = self._buf(tag + ".x16", (M + 1) * KP, h)          # + the padding token's row
            Mp = desc.n_seq * (32 if desc.seq_len <= 32 else 64)      # whole 32-row blocks per sequence
            ctx = self._buf(tag + ".ctx16", Mp * DP, h)
            t = self._buf(tag + ".t16", Mp * QP, h) if need_bwd else None
            w = self._buf(tag + ".w", M) if need_bwd else None
            nbytes = int(self.lib.nrms_encoder_fwd_scratch_bytes(C.byref(desc)))
            # one scratch per tag: a training forward's token / title lists stay untouched until its backward reads them
            # (NRMS_FLAG_FWD_SCRATCH_KEPT), whatever inference or user-encoder passes run in between
            scratch = self._buf(tag + ".scratch16", (nbytes + 3) // 4)
            attn = self._buf(tag + ".attn16", Mp * DP, h) if desc.use_output_proj else None   # head concatenation (fused16_v1.hip)
            dp = lambda z: None if z is None else z.data_ptr()
            return _lib.EncoderActs(x=dp(x), qkv=None, attn=dp(attn), ctx=dp(ctx), t=dp(t), w=dp(w), scratch=dp(scratch))
        x = self._buf(tag + ".x", M * d) if gather else None
        if desc is not None and desc.flags & _lib.NRMS_FLAG_FUSED_SEQ64:
            # acts.qkv = operand fragments of the fused user-encoder kernels (include/nrms_hip.h, NRMS_FLAG_FUSED_SEQ64)
            nb = int(self.lib.nrms_encoder_fused_qkv_bytes(C.byref(desc)))
            qkv = self._buf(tag + ".qkvf", (nb + 3) // 4)
            ctx = self._buf(tag + ".ctx", M * d) if need_bwd else None
            t = self._buf(tag + ".t", M * q) if need_bwd else None
            w = self._buf(tag + ".w", M) if need_bwd else None
            ns = int(self.lib.nrms_encoder_fwd_scratch_bytes(C.byref(desc)))
            scratch = self._buf("fwd_scratch64", (ns + 3) // 4)
            dp = lambda z: None if z is None else z.data_ptr()
            return _lib.EncoderActs(x=None, qkv=dp(qkv), attn=None, ctx=dp(ctx), t=dp(t), w=dp(w), scratch=dp(scratch))
        qkv = self._buf(tag + ".qkv", M * 3 * d)
        attn = self._buf(tag + ".attn", M * d) if self.dims.output_proj else None
        ctx = self._buf(tag + ".ctx", M * d)
        t = self._buf(tag + ".t", M * q) if need_bwd else None
        w = self._buf(tag + ".w", M) if need_bwd else None
        # head-major W_qkv / b_qkv copies + bf16 weight planes: an upper bound of nrms_encoder_fwd_scratch_bytes
        scratch = self._buf("fwd_scratch", 3 * d * d + 3 * d + 256 + (3 * d + 32) * (d + 32) + (M + M // 1024 + 512 if gather else 0))
        dp = lambda z: None if z is None else z.data_ptr()
        return _lib.EncoderActs(x=dp(x), qkv=dp(qkv), attn=dp(attn), ctx=dp(ctx), t=dp(t), w=dp(w), scratch=dp(scratch))

    # ---- word-id validation (nn.Embedding raises on an out-of-range index; here: device-side, deferred) ----
    def sanitize_ids(self, src, dst):
        """dst = src with ids outside [0, n_words) replaced by the padding id 0 (so no kernel can index out of
        bounds), counting them on the device.  The count reaches the host asynchronously: poll_ids() on a later
        call, or check_ids() (synchronises), raises NrmsError."""
        n = src.numel()
        if n == 0:
            return dst
        fn = self.lib.nrms_sanitize_ids_i32 if src.dtype == torch.int32 else self.lib.nrms_sanitize_ids
        rc = fn(_lib.ptr(src), _lib.ptr(dst), C.c_int64(n), int(self.dims.n_words), _lib.ptr(self._bad_ids), _stream())
        _lib.check(rc, "nrms_sanitize_ids")
        self._bad_host.copy_(self._bad_ids, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._bad_event = ev
        return dst

    def _raise_bad_ids(self):
        n = int(self._bad_host.item())
        if n:
            self._bad_ids.zero_()
            self._bad_host.zero_()
            raise _lib.NrmsError("%d word id(s) outside [0, %d) reached the embedding gather (treated as the padding "
                                 "id on the device; the vocabulary and the embedding table disagree)" % (n, self.dims.n_words))

    def poll_ids(self):
        """Non-blocking: raises if an earlier call's id check has completed and found out-of-range ids."""
        if self._bad_event is not None and self._bad_event.query():
            self._bad_event = None
            self._raise_bad_ids()

    def check_ids(self):
        """Blocking form of poll_ids(): waits for the pending id checks."""
        if self._bad_event is not None:
            self._bad_event.synchronize()
            self._bad_event = None
        self._raise_bad_ids()

    # ---- news vectors for an arbitrary list of titles (a-5, a-9 get_news_vector) ---------
    def encode_titles(self, flat, ids, out=None, p_embed=0.0, p_ctx=0.0, seed=0, save=False, tag=None,
                      chunk_titles=32768, mask=None, mask_mode=0, trusted_ids=False):
        """ids [N, L] int64 on the device -> news vectors [N, d].  With save=True (training) the
        activations are kept for the backward and the call is not chunked.  mask [N, L] uint8 with
        mask_mode bits (1 = attention pairs, 2 = pooling) gives nrms_v1's masked primitives."""
        N, L = ids.shape
        d = self.dims.word_embed_size
        if tag is None:
            tag = "news" if save else "news_eval"
        if out is None:
            out = torch.empty(N, d, dtype=torch.float32, device=self.device)
        if trusted_ids:
            ids = ids.contiguous()
        else:
            self.poll_ids()
            ids = self.sanitize_ids(ids.contiguous(), self._buf(tag + ".ids", N * L, torch.int64)[:N * L].view(N, L))
        if mask is not None:
            mask = mask.contiguous()
        w = self._weights(flat, "news_encoder")
        step = N if save else min(N, chunk_titles)
        for s0 in range(0, N, max(step, 1)):
            n = min(step, N - s0)
            desc = self._desc("news_encoder", n, L, p_embed, p_ctx, seed, mask_mode if mask is not None else 0, training=save)
            acts = self._acts(tag, n * L, save, gather=True, desc=desc)
            mp = None if mask is None else C.c_void_p(mask[s0:].data_ptr())
            rc = self.lib.nrms_encoder_fwd(C.byref(desc), C.byref(w), C.c_void_p(ids[s0:].data_ptr()), None, mp,
                                           C.byref(acts), C.c_void_p(out[s0:].data_ptr()), _stream())
            _lib.check(rc, "nrms_encoder_fwd(news)")
        return out

    def encode_users(self, flat, news_vectors, out=None, save=False, tag=None, mask=None, mask_mode=0):
        """news_vectors [B, H, d] -> user vectors [B, d] (a-6, a-9 get_user_vector)."""
        B, H, d = news_vectors.shape
        if tag is None:
            tag = "user" if save else "user_eval"
        if out is None:
            out = torch.empty(B, d, dtype=torch.float32, device=self.device)
        w = self._weights(flat, "user_encoder")
        desc = self._desc("user_encoder", B, H, mask_mode=mask_mode if mask is not None else 0, training=save)
        acts = self._acts(tag, B * H, save, desc=desc)
        rc = self.lib.nrms_encoder_fwd(C.byref(desc), C.byref(w), None, C.c_void_p(news_vectors.data_ptr()),
                                       _lib.ptr(None if mask is None else mask.contiguous()), C.byref(acts),
                                       C.c_void_p(out.data_ptr()), _stream())
        _lib.check(rc, "nrms_encoder_fwd(user)")
        return out

    def click_scores(self, cand_vec, user_vec, cand_mask=None, out=None):
        B, Cn, d = cand_vec.shape
        if out is None:
            out = torch.empty(B, Cn, dtype=torch.float32, device=self.device)
        rc = self.lib.nrms_click_score_fwd(B, Cn, d, _lib.ptr(cand_vec), _lib.ptr(user_vec), _lib.ptr(cand_mask),
                                           _lib.ptr(out), _stream())
        _lib.check(rc, "nrms_click_score_fwd")
        return out

    # ---- full model forward (a-8) -------------------------------------------------------
    def forward(self, flat, hist_ids, cand_ids, cand_mask, training, p_drop=0.0, seed=0, user_mask=None,
                user_mask_mode=0):
        """hist_ids [B,H,L], cand_ids [B,C,L] int64 and cand_mask [B,C] uint8 (or None) on the
        device -> scores [B,C].  training=True keeps what backward() needs; p_drop is applied as given
        (the caller passes 0 in eval mode; a train-mode forward under no_grad still drops, as nn.Dropout does)."""
        B, H, L = hist_ids.shape
        Cn = cand_ids.shape[1]
        d = self.dims.word_embed_size
        N = B * (H + Cn)
        # training activations and inference scratch live under different tags: an inference call (get_news_vector,
        # an eval forward) between a training forward and its backward must not overwrite what the backward reads
        sfx = "" if training else "_eval"
        self.poll_ids()
        ids = self._buf("ids" + sfx, N * L, torch.int64)[:N * L].view(N, L)
        self.sanitize_ids(hist_ids.reshape(B * H, L).contiguous(), ids[:B * H])
        self.sanitize_ids(cand_ids.reshape(B * Cn, L).contiguous(), ids[B * H:])
        nv = self._buf("news_vec" + sfx, N * d)[:N * d].view(N, d)
        p = float(p_drop)
        p_embed = 0.0 if self.dims.style == "v1" else p        # nrms_v1 has no embedding dropout (nrms_v1.py:159-161)
        self.encode_titles(flat, ids, out=nv, p_embed=p_embed, p_ctx=p, seed=seed, save=training, tag="news" + sfx,
                           trusted_ids=True)
        hist = nv[:B * H].view(B, H, d)
        cand = nv[B * H:].view(B, Cn, d)
        user = self._buf("user_vec" + sfx, B * d)[:B * d].view(B, d)
        self.encode_users(flat, hist, out=user, save=training, tag="user" + sfx, mask=user_mask, mask_mode=user_mask_mode)
        if cand_mask is not None:
            cand_mask = cand_mask.contiguous()
        scores = torch.empty(B, Cn, dtype=torch.float32, device=self.device)
        self.click_scores(cand, user, cand_mask, out=scores)
        if training:
            self._gen += 1
            self._saved = dict(B=B, H=H, C=Cn, L=L, ids=ids, nv=nv, user=user, mask=cand_mask, p=p, p_embed=p_embed,
                               seed=seed, user_mask=user_mask, user_mask_mode=user_mask_mode if user_mask is not None else 0,
                               gen=self._gen)
        return scores

    def ce_loss(self, scores, grad_scale=None, want_grad=True):
        """Sum over the batch of -log_softmax(scores)[:,0] (device scalar) and, optionally,
        dscores = (softmax - onehot0) * grad_scale."""
        B, Cn = scores.shape
        loss_sum = torch.zeros(1, dtype=torch.float32, device=self.device)
        dscores = torch.empty_like(scores) if want_grad else None
        gs = (1.0 / B) if grad_scale is None else grad_scale
        rc = self.lib.nrms_ce_loss_fwd_bwd(B, Cn, _lib.ptr(scores), _lib.ptr(loss_sum), _lib.ptr(dscores),
                                           C.c_float(gs), _stream())
        _lib.check(rc, "nrms_ce_loss_fwd_bwd")
        return loss_sum, dscores

    def _bwd_workspace(self, *descs):
        nbytes = max(self.lib.nrms_encoder_bwd_workspace_bytes(C.byref(d)) for d in descs)
        return self._buf("bwd_ws", (nbytes + 3) // 4)

    def encode_users_backward(self, flat, gflat, news_vectors, dout, dx=None, tag="user", mask=None, mask_mode=0):
        """Backward of encode_users(save=True): accumulates the user-encoder parameter gradients into
        gflat and returns d(news_vectors) [B, H, d]."""
        B, H, d = news_vectors.shape
        if dx is None:
            dx = torch.empty(B * H, d, dtype=torch.float32, device=self.device)
        desc = self._desc("user_encoder", B, H, mask_mode=mask_mode if mask is not None else 0, training=True)
        ws = self._bwd_workspace(desc)
        w, g = self._weights(flat, "user_encoder"), self._grads(gflat, "user_encoder")
        acts = self._acts(tag, B * H, True, desc=desc)
        rc = self.lib.nrms_encoder_bwd(C.byref(desc), C.byref(w), None, _lib.ptr(news_vectors),
                                       _lib.ptr(None if mask is None else mask.contiguous()), C.byref(acts),
                                       _lib.ptr(dout.contiguous()), C.byref(g), _lib.ptr(dx), _lib.ptr(ws),
                                       C.c_size_t(ws.numel() * 4), _stream())
        _lib.check(rc, "nrms_encoder_bwd(user)")
        return dx[:B * H].view(B, H, d)

    # ---- full model backward ------------------------------------------------------------
    def backward(self, flat, gflat, dscores, table_grad_ready=None, gen=None):
        """Accumulates d(loss)/d(params) into gflat (same layout as flat) given dscores [B,C].
        gen: the generation stamp of the training forward this backward belongs to (saved activations are one
        slot: a later training forward replaces them, and a backward for the earlier one must not run on them).

        table_grad_ready: optional callable invoked as soon as the embedding-table gradient (95 % of the
        gradient bytes) is complete on the stream; the news encoder's d(W_qkv) GEMM is then deferred behind
        it (NRMS_FLAG_DEFER_WQKV) so that a data-parallel caller can start the table all-reduce underneath."""
        sv = self._saved
        if sv is None:
            raise _lib.NrmsError("backward() without a training forward()")
        if gen is not None and gen != sv["gen"]:
            raise _lib.NrmsError("backward() for training forward #%d, but the saved activations belong to forward #%d: "
                                 "a later training forward replaced them (one forward/backward pair at a time)" % (gen, sv["gen"]))
        B, H, Cn, L = sv["B"], sv["H"], sv["C"], sv["L"]
        d = self.dims.word_embed_size
        N = B * (H + Cn)
        # fp16 mode: the loss scale is derived on the device from the gradient each encoder receives (any loss reduction,
        # batch size or world size); a fixed value only for experiments (tools/fp16_grad_stats.py)
        self.poll_grad_overflow()
        self.loss_scale = float(getattr(self, "loss_scale_override", None) or -float(self.loss_scale_backoff))
        nv, user = sv["nv"], sv["user"]
        hist = nv[:B * H]
        cand = nv[B * H:]
        dnv = self._buf("d_news_vec", N * d)[:N * d].view(N, d)
        duser = self._buf("d_user_vec", B * d)[:B * d].view(B, d)
        dscores = dscores.contiguous()
        rc = self.lib.nrms_click_score_bwd(B, Cn, d, _lib.ptr(cand), _lib.ptr(user), _lib.ptr(sv["mask"]),
                                           _lib.ptr(dscores), C.c_void_p(dnv[B * H:].data_ptr()), _lib.ptr(duser),
                                           _stream())
        _lib.check(rc, "nrms_click_score_bwd")
        # user encoder: its input gradient lands directly in the history rows of d(news vectors)
        desc_u = self._desc("user_encoder", B, H, mask_mode=sv["user_mask_mode"], training=True)
        desc_n = self._desc("news_encoder", N, L, sv["p_embed"], sv["p"], sv["seed"], training=True)
        ws = self._bwd_workspace(desc_u, desc_n)
        self.encode_users_backward(flat, gflat, hist.view(B, H, d), duser, dx=dnv, tag="user", mask=sv["user_mask"],
                                   mask_mode=sv["user_mask_mode"])
        wn, gn = self._weights(flat, "news_encoder"), self._grads(gflat, "news_encoder")
        acts_n = self._acts("news", N * L, True, gather=True, desc=desc_n)
        if desc_n.precision == _lib.NRMS_PRECISION_FP16:
            desc_n.flags |= _lib.NRMS_FLAG_FWD_SCRATCH_KEPT     # acts_n.scratch is this step's forward scratch (its own buffer)
        if table_grad_ready is not None:
            desc_n.flags |= _lib.NRMS_FLAG_DEFER_WQKV
        rc = self.lib.nrms_encoder_bwd(C.byref(desc_n), C.byref(wn), _lib.ptr(sv["ids"]), None, None, C.byref(acts_n),
                                       _lib.ptr(dnv), C.byref(gn), None, _lib.ptr(ws),
                                       C.c_size_t(ws.numel() * 4), _stream())
        _lib.check(rc, "nrms_encoder_bwd(news)")
        if table_grad_ready is not None:
            table_grad_ready()
            rc = self.lib.nrms_encoder_bwd_wqkv(C.byref(desc_n), _lib.ptr(sv["ids"]), None, C.byref(acts_n), C.byref(gn),
                                                _lib.ptr(ws), C.c_size_t(ws.numel() * 4), _stream())
            _lib.check(rc, "nrms_encoder_bwd_wqkv(news)")

    def adam_step(self, flat, gflat, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                  grad_scale=1.0, guard=None):
        """torch.optim.Adam's update on flat buffers.  guard (default: on in the fp16 mode): elements with a non-finite
        gradient are left out of the update and counted (nrms_adam_step_guarded); note_grad_check() afterwards lets the
        count reach the host without a sync."""
        if guard is None:
            guard = self.precision == "fp16"
        if guard:
            rc = self.lib.nrms_adam_step_guarded(C.c_size_t(flat.numel()), _lib.ptr(flat), _lib.ptr(gflat), _lib.ptr(exp_avg),
                                                 _lib.ptr(exp_avg_sq), C.c_double(lr), C.c_double(betas[0]), C.c_double(betas[1]),
                                                 C.c_double(eps), int(step), C.c_float(grad_scale), _lib.ptr(self._grad_bad), _stream())
            _lib.check(rc, "nrms_adam_step_guarded")
            return
        rc = self.lib.nrms_adam_step(C.c_size_t(flat.numel()), _lib.ptr(flat), _lib.ptr(gflat), _lib.ptr(exp_avg),
                                     _lib.ptr(exp_avg_sq), C.c_double(lr), C.c_double(betas[0]), C.c_double(betas[1]),
                                     C.c_double(eps), int(step), C.c_float(grad_scale), _stream())
        _lib.check(rc, "nrms_adam_step")

    # ---- fp16 gradient overflow: device-side count, host-side back-off of the loss scale, no sync ----------------------
    def grad_guard(self, gflat):
        """inf / nan elements of a gradient buffer -> 0, counted (for callers that run their own optimizer: the autograd path)."""
        rc = self.lib.nrms_grad_guard(C.c_size_t(gflat.numel()), _lib.ptr(gflat), _lib.ptr(self._grad_bad), _stream())
        _lib.check(rc, "nrms_grad_guard")

    def note_grad_check(self):
        """Behind the guarded optimizer / grad_guard of a step: start the asynchronous read-back of the counter."""
        if self._grad_bad_event is not None and not self._grad_bad_event.query():
            return                                       # the previous read-back is still in flight: its count covers this step too
        self._grad_bad_host.copy_(self._grad_bad, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._grad_bad_event = ev

    def poll_grad_overflow(self, block=False):
        """Non-blocking (unless block): if a finished read-back shows non-finite gradient elements, give the fp16 backward two
        more powers of two of head room from the next step on (and warn once per event); after 2 000 clean checks take one back."""
        ev = self._grad_bad_event
        if ev is None:
            return self.loss_scale_backoff
        if block:
            ev.synchronize()
        elif not ev.query():
            return self.loss_scale_backoff
        self._grad_bad_event = None
        n = int(self._grad_bad_host.item())
        if n:
            self._grad_bad.zero_()
            self._grad_bad_host.zero_()
            self.grad_overflow_steps += 1
            self._clean_polls = 0
            if self.loss_scale_backoff < 24:
                self.loss_scale_backoff = min(24, self.loss_scale_backoff + 2)
            import warnings
            warnings.warn("NRMS fp16 backward: %d non-finite gradient element(s) were skipped by the optimizer (an fp16 overflow of "
                          "the loss-scaled gradient tensors, or an inf / nan loss); loss-scale head room raised to 2^%d below the "
                          "default" % (n, self.loss_scale_backoff), RuntimeWarning, stacklevel=2)
        else:
            self._clean_polls += 1
            if self._clean_polls >= 2000 and self.loss_scale_backoff > 0:
                self.loss_scale_backoff -= 1
                self._clean_polls = 0
        return self.loss_scale_backoff

    def impression_auc(self, scores, labels, lens):
        """scores [n, Cmax] fp32, labels [n, Cmax] uint8, lens [n] int32 (device) -> float64 AUC per impression."""
        n, cmax = scores.shape
        auc = torch.empty(n, dtype=torch.float64, device=self.device)
        rc = self.lib.nrms_impression_auc(n, cmax, _lib.ptr(scores.contiguous()), _lib.ptr(labels.contiguous()),
                                          _lib.ptr(lens.contiguous()), _lib.ptr(auc), _stream())
        _lib.check(rc, "nrms_impression_auc")
        return auc

    # ---- inference with unique-title caching (SURVEY f-1) -------------------------------------
    def group_rows(self, rows):
        """rows [N, L] int64 (device) -> (inverse [N] int32, rep_rows [U] int32, U): exact grouping of equal rows
        by a device hash table (nrms_title_dedup): rows[rep_rows[inverse[t]]] == rows[t].  One host read-back (U)."""
        N, L = rows.shape
        size = 1 << max(int(2 * N - 1).bit_length(), 4)
        table = self._buf("dedup_table", size, torch.int32)
        inverse = torch.empty(N, dtype=torch.int32, device=self.device)
        rep = self._buf("dedup_rep", N, torch.int32)
        n_unique = self._buf("dedup_count", 1, torch.int32)
        rc = self.lib.nrms_title_dedup(_lib.ptr(rows), C.c_int64(N), int(L), _lib.ptr(table), C.c_int64(size),
                                       _lib.ptr(inverse), _lib.ptr(rep), _lib.ptr(n_unique), _stream())
        _lib.check(rc, "nrms_title_dedup")
        U = int(n_unique.item())
        return inverse, rep[:U], U

    def unique_titles(self, ids):
        """ids [N, L] -> (one title per group [U, L], inverse [N] int32) with ids == uniq[inverse]."""
        inverse, rep, _ = self.group_rows(ids)
        return ids.index_select(0, rep), inverse

    def click_scores_indexed(self, news_vec, index, user_vec, B, Cn, cand_mask=None):
        """scores [B, Cn] with candidate slot (b, c) = row index[b*Cn + c] of news_vec [n, d]."""
        out = torch.empty(B, Cn, dtype=torch.float32, device=self.device)
        rc = self.lib.nrms_click_score_indexed(B, Cn, news_vec.shape[1], _lib.ptr(news_vec), C.c_int64(news_vec.shape[0]),
                                               _lib.ptr(index), _lib.ptr(user_vec), _lib.ptr(cand_mask), _lib.ptr(out),
                                               _stream())
        _lib.check(rc, "nrms_click_score_indexed")
        return out

    def forward_dedup(self, flat, hist_ids, cand_ids, cand_mask):
        """Same scores as forward(training=False), but every distinct title of the batch is encoded
        once (the reference encodes all B*(H+C) slots, 350 per user at C=300, most of them padding
        or repeats: train_eval.py:229-273 / data_handler.py:174-177); candidate vectors are read by index."""
        B, H, L = hist_ids.shape
        Cn = cand_ids.shape[1]
        d = self.dims.word_embed_size
        N = B * (H + Cn)
        self.poll_ids()
        ids = self._buf("ids_eval", N * L, torch.int64)[:N * L].view(N, L)
        self.sanitize_ids(hist_ids.reshape(B * H, L).contiguous(), ids[:B * H])
        self.sanitize_ids(cand_ids.reshape(B * Cn, L).contiguous(), ids[B * H:])
        uniq, inverse = self.unique_titles(ids)
        vec = self.encode_titles(flat, uniq, tag="news_eval", trusted_ids=True)
        hist = vec.index_select(0, inverse[:B * H]).view(B, H, d)
        user = self.encode_users(flat, hist, tag="user_eval")
        if cand_mask is not None:
            cand_mask = cand_mask.contiguous()
        return self.click_scores_indexed(vec, inverse[B * H:], user, B, Cn, cand_mask), int(uniq.shape[0])

    # persistent news-vector cache across evaluation batches, keyed by the news ids the batch dict carries
    # (browsed_ids / candidate_ids, data_handler.py:204-222; 0 = padding slot = all-padding title)
    def news_cache_begin(self, capacity=1 << 14):
        d = self.dims.word_embed_size
        old = getattr(self, "_news_cache_store", None)
        if old is not None and old[0].shape[1] == d:                  # buffers are kept between evaluations; only the
            vec, have = old                                           # validity bits are cleared (weights changed)
            have.zero_()
        else:
            vec = torch.empty(capacity, d, dtype=torch.float32, device=self.device)
            have = torch.zeros(capacity, dtype=torch.bool, device=self.device)
        self._news_cache = dict(vec=vec, have=have, encoded=0, hits=0)

    def news_cache_end(self):
        st = getattr(self, "_news_cache", None)
        self._news_cache = None
        if st is None:
            return None
        self._news_cache_store = (st["vec"], st["have"])
        return dict(encoded=st["encoded"], lookups=st["hits"])

    def forward_cached(self, flat, hist_ids, cand_ids, hist_news, cand_news, cand_mask):
        """forward(training=False) for batches that carry news ids: a news item is encoded the first time it is
        seen during the evaluation (weights are constant between news_cache_begin / news_cache_end) and looked up
        afterwards -- the offline news-vector caching get_news_vector exists for (nrms_v0.py:278-289)."""
        st = self._news_cache
        B, H, L = hist_ids.shape
        Cn = cand_ids.shape[1]
        d = self.dims.word_embed_size
        N = B * (H + Cn)
        news = torch.cat([hist_news.reshape(-1), cand_news.reshape(-1)]).to(torch.int64).contiguous()
        inverse, rep, _ = self.group_rows(news.view(N, 1))
        u = news.index_select(0, rep)                                  # the distinct news ids of the batch
        lo, top = (int(v) for v in torch.stack([u.min(), u.max()]).tolist()) if N else (0, 0)
        if lo < 0:
            raise _lib.NrmsError("negative news id in browsed_ids / candidate_ids")
        if top + 1 > st["vec"].shape[0]:                              # grow (amortised doubling)
            cap = max(top + 1, 2 * st["vec"].shape[0])
            vec = torch.empty(cap, d, dtype=torch.float32, device=self.device)
            have = torch.zeros(cap, dtype=torch.bool, device=self.device)
            vec[:st["vec"].shape[0]] = st["vec"]
            have[:st["have"].shape[0]] = st["have"]
            st["vec"], st["have"] = vec, have
        need = ~st["have"].index_select(0, u)
        rows = rep[need].to(torch.int64)                               # a slot of the batch holding each news item not cached yet
        if rows.numel():
            missing = u[need]
            titles = torch.where((rows < B * H).unsqueeze(1),
                                 hist_ids.reshape(B * H, L).index_select(0, rows.clamp(max=B * H - 1)),
                                 cand_ids.reshape(B * Cn, L).index_select(0, (rows - B * H).clamp(min=0)))
            st["vec"].index_copy_(0, missing, self.encode_titles(flat, titles.contiguous(), tag="news_eval"))
            st["have"][missing] = True
            st["encoded"] += int(missing.numel())
        st["hits"] += N
        hist = st["vec"].index_select(0, news[:B * H]).view(B, H, d)
        user = self.encode_users(flat, hist, tag="user_eval")
        if cand_mask is not None:
            cand_mask = cand_mask.contiguous()
        return self.click_scores_indexed(st["vec"], news[B * H:].to(torch.int32), user, B, Cn, cand_mask)

    def dropout_keep_mask(self, seed, site, n_rows, p_drop, d=None, fp16_ctx=False):
        """Keep mask of a dropout site over [n_rows, d] (d defaults to the model width).  fp16_ctx: the context dropout
        of the fp16 mode -- padded width 320 (32 columns per head), 16-bit-field scheme, the two middle index bits of a
        column inside its head block swapped (include/nrms_hip.h, NRMS_DROPOUT_FIELDS16); returned in natural column order."""
        d = self.dims.word_embed_size if d is None else int(d)
        if fp16_ctx:
            d, site = _lib.NRMS_FP16_DP, site | _lib.NRMS_DROPOUT_FIELDS16
        keep = torch.empty(n_rows * d, dtype=torch.uint8, device=self.device)
        rc = self.lib.nrms_dropout_keep_mask(C.c_uint64(seed), site, C.c_int64(n_rows), d, C.c_float(p_drop),
                                             _lib.ptr(keep), _stream())
        _lib.check(rc, "nrms_dropout_keep_mask")
        keep = keep.view(n_rows, d)
        if fp16_ctx:
            col = torch.arange(d, device=self.device)
            a, b, c, e = col >> 4, (col >> 3) & 1, (col >> 2) & 1, col & 3
            keep = keep.index_select(1, 16 * a + 8 * c + 4 * b + e)
        return keep

    # ---- timing (bench.py roofline leg) --------------------------------------------------
    def timing(self, enable: bool):
        self.lib.nrms_timing_enable(1 if enable else 0)

    def timing_reset(self):
        self.lib.nrms_timing_reset()

    def timing_read(self, prefix: str):
        ms, n = C.c_double(0.0), C.c_int64(0)
        _lib.check(self.lib.nrms_timing_read(prefix.encode(), C.byref(ms), C.byref(n)), "nrms_timing_read")
        return ms.value, n.value
