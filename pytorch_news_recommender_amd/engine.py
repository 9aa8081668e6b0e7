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
from .synth import ENCODERS, param_names


@dataclass(frozen=True)
class ModelDims:
    n_words: int
    word_embed_size: int
    num_attention_heads: int
    query_vector_dim: int


class FlatLayout:
    """Offsets (in floats) of the 19 reference-named tensors inside the flat buffer.
    W_Q/W_K/W_V (and their biases) are adjacent so the kernels see one [3d, d] matrix."""

    def __init__(self, dims: ModelDims):
        V, d, q = dims.n_words, dims.word_embed_size, dims.query_vector_dim
        if d % 4 or q % 4:
            raise ValueError("word_embed_size and query_vector_dim must be multiples of 4")
        self.dims = dims
        self.entries = {}
        off = 0
        for name in param_names():
            if name.endswith("word_embedding.0.weight"):
                shp = (V, d)
            elif name.endswith("attention_query_vector") or name.endswith("linear.bias"):
                shp = (q,)
            elif name.endswith("linear.weight"):
                shp = (q, d)
            elif name.endswith(".bias"):
                shp = (d,)
            else:
                shp = (d, d)
            n = 1
            for s in shp:
                n *= s
            self.entries[name] = (off, shp, n)
            off += n
        self.total = off
        self.blocks = {}
        for enc in ENCODERS:
            a = enc + ".multihead_self_attention."
            b = enc + ".additive_attention."
            self.blocks[enc] = {
                "w_qkv": self.entries[a + "W_Q.weight"][0],
                "b_qkv": self.entries[a + "W_Q.bias"][0],
                "w_add": self.entries[b + "linear.weight"][0],
                "b_add": self.entries[b + "linear.bias"][0],
                "q_vec": self.entries[b + "attention_query_vector"][0],
            }
            # adjacency the kernels rely on
            assert self.entries[a + "W_K.weight"][0] == self.blocks[enc]["w_qkv"] + d * d
            assert self.entries[a + "W_V.weight"][0] == self.blocks[enc]["w_qkv"] + 2 * d * d
            assert self.entries[a + "W_K.bias"][0] == self.blocks[enc]["b_qkv"] + d
            assert self.entries[a + "W_V.bias"][0] == self.blocks[enc]["b_qkv"] + 2 * d
        self.table = self.entries["news_encoder.word_embedding.0.weight"][0]

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shp, n = self.entries[name]
        return flat[off:off + n].view(shp)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class NRMSEngine:
    """One NRMS forward / backward / optimizer step on one GPU."""

    def __init__(self, dims: ModelDims, device):
        self.lib = _lib.load()
        self.dims = dims
        self.layout = FlatLayout(dims)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NrmsError("the NRMS HIP engine needs a GPU device (got %s); there is no CPU path" % device)
        self._bufs = {}
        self._saved = None

    # ---- buffers ----------------------------------------------------------------------
    def _buf(self, key, numel, dtype=torch.float32):
        t = self._bufs.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(max(int(numel), 1), dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def _desc(self, n_seq, seq_len, vocab, p_drop, seed):
        d = self.dims
        return _lib.EncoderDesc(n_seq=n_seq, seq_len=seq_len, d_model=d.word_embed_size,
                                n_heads=d.num_attention_heads, q_dim=d.query_vector_dim, vocab=vocab,
                                p_drop=float(p_drop), precision=_lib.NRMS_PRECISION_FP32, seed=int(seed))

    def _weights(self, flat, enc):
        b = self.layout.blocks[enc]
        base = flat.data_ptr()
        table = base + 4 * self.layout.table if enc == "news_encoder" else None
        return _lib.EncoderWeights(table=table, w_qkv=base + 4 * b["w_qkv"], b_qkv=base + 4 * b["b_qkv"],
                                   w_add=base + 4 * b["w_add"], b_add=base + 4 * b["b_add"],
                                   q_vec=base + 4 * b["q_vec"])

    def _grads(self, gflat, enc):
        b = self.layout.blocks[enc]
        base = gflat.data_ptr()
        table = base + 4 * self.layout.table if enc == "news_encoder" else None
        return _lib.EncoderGrads(table=table, w_qkv=base + 4 * b["w_qkv"], b_qkv=base + 4 * b["b_qkv"],
                                 w_add=base + 4 * b["w_add"], b_add=base + 4 * b["b_add"],
                                 q_vec=base + 4 * b["q_vec"])

    def _acts(self, tag, M, need_bwd, gather=False):
        d, q = self.dims.word_embed_size, self.dims.query_vector_dim
        x = self._buf(tag + ".x", M * d) if gather else None
        qkv = self._buf(tag + ".qkv", M * 3 * d)
        ctx = self._buf(tag + ".ctx", M * d)
        t = self._buf(tag + ".t", M * q) if need_bwd else None
        w = self._buf(tag + ".w", M) if need_bwd else None
        acts = _lib.EncoderActs(x=None if x is None else x.data_ptr(), qkv=qkv.data_ptr(), ctx=ctx.data_ptr(),
                                t=None if t is None else t.data_ptr(), w=None if w is None else w.data_ptr())
        return acts

    # ---- news vectors for an arbitrary list of titles (a-5, a-9 get_news_vector) ---------
    def encode_titles(self, flat, ids, out=None, p_drop=0.0, seed=0, save=False, tag="news", chunk_titles=32768):
        """ids [N, L] int64 on the device -> news vectors [N, d].  With save=True (training) the
        activations are kept for encode_titles_bwd and the call is not chunked."""
        N, L = ids.shape
        d = self.dims.word_embed_size
        if out is None:
            out = torch.empty(N, d, dtype=torch.float32, device=self.device)
        ids = ids.contiguous()
        w = self._weights(flat, "news_encoder")
        step = N if save else min(N, chunk_titles)
        for s0 in range(0, N, max(step, 1)):
            n = min(step, N - s0)
            desc = self._desc(n, L, self.dims.n_words, p_drop, seed)
            acts = self._acts(tag, n * L, save, gather=True)
            rc = self.lib.nrms_encoder_fwd(C.byref(desc), C.byref(w), C.c_void_p(ids[s0:].data_ptr()), None,
                                           C.byref(acts), C.c_void_p(out[s0:].data_ptr()), _stream())
            _lib.check(rc, "nrms_encoder_fwd(news)")
        return out

    def encode_users(self, flat, news_vectors, out=None, save=False, tag="user"):
        """news_vectors [B, H, d] -> user vectors [B, d] (a-6, a-9 get_user_vector)."""
        B, H, d = news_vectors.shape
        if out is None:
            out = torch.empty(B, d, dtype=torch.float32, device=self.device)
        w = self._weights(flat, "user_encoder")
        desc = self._desc(B, H, 0, 0.0, 0)
        acts = self._acts(tag, B * H, save)
        rc = self.lib.nrms_encoder_fwd(C.byref(desc), C.byref(w), None, C.c_void_p(news_vectors.data_ptr()),
                                       C.byref(acts), C.c_void_p(out.data_ptr()), _stream())
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
    def forward(self, flat, hist_ids, cand_ids, cand_mask, training, p_drop=0.0, seed=0):
        """hist_ids [B,H,L], cand_ids [B,C,L] int64 and cand_mask [B,C] uint8 (or None) on the
        device -> scores [B,C].  training=True keeps what backward() needs."""
        B, H, L = hist_ids.shape
        Cn = cand_ids.shape[1]
        d = self.dims.word_embed_size
        N = B * (H + Cn)
        ids = self._buf("ids", N * L, torch.int64)[:N * L].view(N, L)
        ids[:B * H].copy_(hist_ids.reshape(B * H, L))
        ids[B * H:].copy_(cand_ids.reshape(B * Cn, L))
        nv = self._buf("news_vec", N * d)[:N * d].view(N, d)
        p = p_drop if training else 0.0
        self.encode_titles(flat, ids, out=nv, p_drop=p, seed=seed, save=training)
        hist = nv[:B * H].view(B, H, d)
        cand = nv[B * H:].view(B, Cn, d)
        user = self._buf("user_vec", B * d)[:B * d].view(B, d)
        self.encode_users(flat, hist, out=user, save=training)
        if cand_mask is not None:
            cand_mask = cand_mask.contiguous()
        scores = torch.empty(B, Cn, dtype=torch.float32, device=self.device)
        self.click_scores(cand, user, cand_mask, out=scores)
        if training:
            self._saved = dict(B=B, H=H, C=Cn, L=L, ids=ids, nv=nv, user=user, mask=cand_mask, p=p, seed=seed)
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

    # ---- full model backward ------------------------------------------------------------
    def backward(self, flat, gflat, dscores):
        """Accumulates d(loss)/d(params) into gflat (same layout as flat) given dscores [B,C]."""
        sv = self._saved
        if sv is None:
            raise _lib.NrmsError("backward() without a training forward()")
        B, H, Cn, L = sv["B"], sv["H"], sv["C"], sv["L"]
        d = self.dims.word_embed_size
        N = B * (H + Cn)
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
        desc_u = self._desc(B, H, 0, 0.0, 0)
        desc_n = self._desc(N, L, self.dims.n_words, sv["p"], sv["seed"])
        ws_bytes = max(self.lib.nrms_encoder_bwd_workspace_bytes(C.byref(desc_u)),
                       self.lib.nrms_encoder_bwd_workspace_bytes(C.byref(desc_n)))
        ws = self._buf("bwd_ws", (ws_bytes + 3) // 4)
        wu, gu = self._weights(flat, "user_encoder"), self._grads(gflat, "user_encoder")
        acts_u = self._acts("user", B * H, True)
        rc = self.lib.nrms_encoder_bwd(C.byref(desc_u), C.byref(wu), None, _lib.ptr(hist), C.byref(acts_u),
                                       _lib.ptr(duser), C.byref(gu), _lib.ptr(dnv), _lib.ptr(ws),
                                       C.c_size_t(ws.numel() * 4), _stream())
        _lib.check(rc, "nrms_encoder_bwd(user)")
        wn, gn = self._weights(flat, "news_encoder"), self._grads(gflat, "news_encoder")
        acts_n = self._acts("news", N * L, True, gather=True)
        rc = self.lib.nrms_encoder_bwd(C.byref(desc_n), C.byref(wn), _lib.ptr(sv["ids"]), None, C.byref(acts_n),
                                       _lib.ptr(dnv), C.byref(gn), None, _lib.ptr(ws),
                                       C.c_size_t(ws.numel() * 4), _stream())
        _lib.check(rc, "nrms_encoder_bwd(news)")

    def adam_step(self, flat, gflat, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                  grad_scale=1.0):
        rc = self.lib.nrms_adam_step(C.c_size_t(flat.numel()), _lib.ptr(flat), _lib.ptr(gflat), _lib.ptr(exp_avg),
                                     _lib.ptr(exp_avg_sq), C.c_float(lr), C.c_float(betas[0]), C.c_float(betas[1]),
                                     C.c_float(eps), int(step), C.c_float(grad_scale), _stream())
        _lib.check(rc, "nrms_adam_step")

    def impression_auc(self, scores, labels, lens):
        """scores [n, Cmax] fp32, labels [n, Cmax] uint8, lens [n] int32 (device) -> float64 AUC per impression."""
        n, cmax = scores.shape
        auc = torch.empty(n, dtype=torch.float64, device=self.device)
        rc = self.lib.nrms_impression_auc(n, cmax, _lib.ptr(scores.contiguous()), _lib.ptr(labels.contiguous()),
                                          _lib.ptr(lens.contiguous()), _lib.ptr(auc), _stream())
        _lib.check(rc, "nrms_impression_auc")
        return auc

    # ---- inference with unique-title caching (SURVEY f-1) -------------------------------------
    def forward_dedup(self, flat, hist_ids, cand_ids, cand_mask):
        """Same scores as forward(training=False), but every distinct title of the batch is encoded
        once (the reference encodes all B*(H+C) slots, 350 per user at C=300, most of them padding
        or repeats: train_eval.py:229-273 / data_handler.py:174-177)."""
        B, H, L = hist_ids.shape
        Cn = cand_ids.shape[1]
        d = self.dims.word_embed_size
        ids = torch.cat([hist_ids.reshape(B * H, L), cand_ids.reshape(B * Cn, L)], dim=0)
        uniq, inverse = torch.unique(ids, dim=0, return_inverse=True)
        vec = self.encode_titles(flat, uniq, tag="news_eval")
        nv = vec.index_select(0, inverse)
        hist = nv[:B * H].view(B, H, d).contiguous()
        cand = nv[B * H:].view(B, Cn, d).contiguous()
        user = self.encode_users(flat, hist, tag="user_eval")
        if cand_mask is not None:
            cand_mask = cand_mask.contiguous()
        return self.click_scores(cand, user, cand_mask), int(uniq.shape[0])

    def dropout_keep_mask(self, seed, site, n_rows, p_drop):
        d = self.dims.word_embed_size
        keep = torch.empty(n_rows * d, dtype=torch.uint8, device=self.device)
        rc = self.lib.nrms_dropout_keep_mask(C.c_uint64(seed), site, C.c_int64(n_rows), d, C.c_float(p_drop),
                                             _lib.ptr(keep), _stream())
        _lib.check(rc, "nrms_dropout_keep_mask")
        return keep.view(n_rows, d)

    # ---- timing (bench.py roofline leg) --------------------------------------------------
    def timing(self, enable: bool):
        self.lib.nrms_timing_enable(1 if enable else 0)

    def timing_reset(self):
        self.lib.nrms_timing_reset()

    def timing_read(self, prefix: str):
        ms, n = C.c_double(0.0), C.c_int64(0)
        _lib.check(self.lib.nrms_timing_read(prefix.encode(), C.byref(ms), C.byref(n)), "nrms_timing_read")
        return ms.value, n.value
