"""HieRec-style hierarchical interest model on MI355X (BASELINE configs[3]).

PARITY UNPINNED.  The reference repository holds no implementation of this model (``/root/reference/MIND_2020/model/tanr.py`` is an
empty file, README.md:3 only names Adressa), so there is nothing of the reference's to match; the model is specified here, from
HieRec (Qi et al., ACL 2021), and checked against ``oracle/segpool_oracle.py`` (a torch restatement of this specification built on
the reference's own additive attention, ``model/nrms_v0.py:100-126``).  It keeps the reference's plugin API -- ``Model(config)``,
``forward(batch_dict) -> FloatTensor[B, C]`` with masked candidates at -1e9 -- and reads these keys of the ``MyDataset`` batch
dict (``data_handler.py:185-250``): ``browsed_titles``, ``browsed_categ_ids``, ``browsed_subcateg_ids``, ``browsed_mask``,
``candidate_titles``, ``candidate_categ_ids``, ``candidate_subcateg_ids``, ``candidate_mask``.

Specification (d = word_embed_size, q = query_vector_dim):
  n_k            = NRMS news encoder of a title (``nrms_v0.py:154-176``: the same kernels as ``model/nrms_hip.py``)
  AddPool(X; m)  = sum_k softmax_k(q_m . tanh(W_m x_k + b_m)) x_k                      (additive attention over a set of rows)
  sub-topic level: u1[s] = AddPool({n_k : k clicked, subtopic(k) = s}; sub) + E_sub[s]      for every sub-topic s the user clicked
  topic level    : u2[t] = AddPool({u1[s] : topic(s) = t}; top) + E_top[t]                   (topic(s) = topic of s's first click)
  user level     : ug    = AddPool({u2[t]}; user)                                            (zero if the user has no valid click)
  score(c)       = l_s f_s <n_c, u1[s_c]> + l_t f_t <n_c, u2[t_c]> + (1 - l_s - l_t) <n_c, ug>
                   f_s, f_t = share of the user's clicks in the candidate's sub-topic / topic (terms vanish where the user never
                   clicked); l_s = config.hierec_lambda_sub (0.7), l_t = config.hierec_lambda_top (0.15)
Every aggregation is ``nrms_segment_pool_fwd / _bwd`` over index lists built on the device by ``nrms_hier_tree_build``
(csrc/segpool.hip, csrc/hier.hip); no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from ..engine import ModelDims, NRMSEngine, _stream
from ..segpool import SegmentPool
from . import nrms_hip
from ._flat_model import AdditiveParams, FlatHipModel, FlatLayout2, additive_entries

LEVELS = ("subtopic_attention", "topic_attention", "user_attention")


def hier_layout(dims: ModelDims, n_sub: int, n_top: int):
    """[ table | news encoder (nrms_v0 names) | E_sub | E_top | three additive-attention modules ]"""
    d, q = dims.word_embed_size, dims.query_vector_dim
    extra = [("subtopic_embedding.weight", (n_sub, d)), ("topic_embedding.weight", (n_top, d))]
    for lv in LEVELS:
        extra += additive_entries(lv, q, d)
    return FlatLayout2(dims, extra)


class HieRecEngine(NRMSEngine):
    """Forward / backward of the hierarchical model on one GPU: the NRMS engine's news encoder + the index and aggregation calls."""

    def __init__(self, dims, layout, device, precision, n_sub, n_top, lambda_sub, lambda_top):
        super().__init__(dims, device, precision=precision)
        self.layout = layout
        self.n_sub, self.n_top = int(n_sub), int(n_top)
        self.lambda_sub, self.lambda_top = float(lambda_sub), float(lambda_top)
        self._pools = None

    def _pool_precision(self):
        return "fp32" if self.precision == "fp32" else "bf16x3"

    def _i32(self, key, n):
        return self._buf(key, n, torch.int32)[:n]

    def forward(self, flat, batch, training, p_drop=0.0, seed=0):
        bt, ct = batch["browsed_titles"], batch["candidate_titles"]
        B, H, L = bt.shape
        Cn = ct.shape[1]
        d, q = self.dims.word_embed_size, self.dims.query_vector_dim
        if H > 64:
            raise _lib.NrmsError("hierec: history_len %d > 64" % H)
        N = B * (H + Cn)
        sfx = "" if training else "_eval"
        self.poll_ids()
        ids = self._buf("ids" + sfx, N * L, torch.int64)[:N * L].view(N, L)
        self.sanitize_ids(bt.reshape(B * H, L).contiguous(), ids[:B * H])
        self.sanitize_ids(ct.reshape(B * Cn, L).contiguous(), ids[B * H:])
        nv = self._buf("news_vec" + sfx, N * d)[:N * d].view(N, d)
        p = float(p_drop)
        self.encode_titles(flat, ids, out=nv, p_embed=p, p_ctx=p, seed=seed, save=training, tag="news" + sfx, trusted_ids=True)
        hist, cand = nv[:B * H], nv[B * H:]
        # ---- the interest tree's index lists (device; validated category ids)
        n = B * H
        valid = batch["browsed_mask"].to(torch.uint8).contiguous()
        topic = self._buf("topic" + sfx, n, torch.int64)[:n]
        sub = self._buf("sub" + sfx, n, torch.int64)[:n]
        ctop = self._buf("ctopic" + sfx, B * Cn, torch.int64)[:B * Cn]
        csub = self._buf("csub" + sfx, B * Cn, torch.int64)[:B * Cn]
        for src, dst, vocab in ((batch["browsed_categ_ids"], topic, self.n_top), (batch["browsed_subcateg_ids"], sub, self.n_sub),
                                (batch["candidate_categ_ids"], ctop, self.n_top), (batch["candidate_subcateg_ids"], csub, self.n_sub)):
            s_ = src.reshape(-1).to(torch.int64).contiguous()
            _lib.check(self.lib.nrms_sanitize_ids(_lib.ptr(s_), _lib.ptr(dst), C.c_int64(s_.numel()), int(vocab), _lib.ptr(self._bad_ids),
                                                  _stream()), "nrms_sanitize_ids")
        t = {k: self._i32(k + sfx, m) for k, m in (("l1_ptr", n + 1), ("l1_idx", n), ("l1_sub", n), ("l1_top", n), ("l1_cnt", n),
                                                    ("l2_ptr", n + 1), ("l2_idx", n), ("l2_top", n), ("l2_cnt", n), ("l3_ptr", B + 1),
                                                    ("l3_idx", n), ("n_valid", B), ("sub_slot", B * Cn), ("top_slot", B * Cn))}
        sb = int(self.lib.nrms_hier_tree_scratch_bytes(B, H))
        scratch = self._buf("hier_scratch", (sb + 3) // 4 + 2)
        rc = self.lib.nrms_hier_tree_build(B, H, _lib.ptr(valid), _lib.ptr(topic), _lib.ptr(sub), *[_lib.ptr(t[k]) for k in (
            "l1_ptr", "l1_idx", "l1_sub", "l1_top", "l1_cnt", "l2_ptr", "l2_idx", "l2_top", "l2_cnt", "l3_ptr", "l3_idx", "n_valid")],
            _lib.ptr(scratch), C.c_size_t(scratch.numel() * 4), _stream())
        _lib.check(rc, "nrms_hier_tree_build")
        # ---- three aggregations
        lay = self.layout
        W = {lv: (lay.view(flat, lv + ".linear.weight"), lay.view(flat, lv + ".linear.bias"), lay.view(flat, lv + ".attention_query_vector"))
             for lv in LEVELS}
        prec = self._pool_precision()
        pools = [SegmentPool(d, q, prec, rows_unique=True) for _ in range(3)]
        u1 = pools[0].forward(hist, *W[LEVELS[0]], t["l1_ptr"], t["l1_idx"])
        _lib.check(self.lib.nrms_hier_add_embedding_fwd(C.c_int64(n), d, self.n_sub, _lib.ptr(t["l1_sub"]), _lib.ptr(t["l1_cnt"]),
                                                        _lib.ptr(lay.view(flat, "subtopic_embedding.weight")), _lib.ptr(u1), _stream()), "add_embedding")
        u2 = pools[1].forward(u1, *W[LEVELS[1]], t["l2_ptr"], t["l2_idx"])
        _lib.check(self.lib.nrms_hier_add_embedding_fwd(C.c_int64(n), d, self.n_top, _lib.ptr(t["l2_top"]), _lib.ptr(t["l2_cnt"]),
                                                        _lib.ptr(lay.view(flat, "topic_embedding.weight")), _lib.ptr(u2), _stream()), "add_embedding")
        ug = pools[2].forward(u2, *W[LEVELS[2]], t["l3_ptr"], t["l3_idx"])
        # ---- hierarchical matching
        sub_frac = self._buf("sub_frac" + sfx, B * Cn)[:B * Cn]
        top_frac = self._buf("top_frac" + sfx, B * Cn)[:B * Cn]
        rc = self.lib.nrms_hier_match(B, Cn, H, _lib.ptr(ctop), _lib.ptr(csub), _lib.ptr(t["l1_sub"]), _lib.ptr(t["l1_cnt"]), _lib.ptr(t["l2_top"]),
                                      _lib.ptr(t["l2_cnt"]), _lib.ptr(t["n_valid"]), _lib.ptr(t["sub_slot"]), _lib.ptr(sub_frac),
                                      _lib.ptr(t["top_slot"]), _lib.ptr(top_frac), _stream())
        _lib.check(rc, "nrms_hier_match")
        mask = batch.get("candidate_mask")
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        scores = torch.empty(B, Cn, dtype=torch.float32, device=self.device)
        rc = self.lib.nrms_hier_score_fwd(B, Cn, d, _lib.ptr(cand), _lib.ptr(u1), _lib.ptr(u2), _lib.ptr(ug), _lib.ptr(t["sub_slot"]),
                                          _lib.ptr(sub_frac), _lib.ptr(t["top_slot"]), _lib.ptr(top_frac), _lib.ptr(mask),
                                          C.c_float(self.lambda_sub), C.c_float(self.lambda_top), _lib.ptr(scores), _stream())
        _lib.check(rc, "nrms_hier_score_fwd")
        self._bad_host.copy_(self._bad_ids, non_blocking=True)
        if training:
            self._gen += 1
            self._saved = dict(B=B, H=H, C=Cn, L=L, ids=ids, nv=nv, u1=u1, u2=u2, ug=ug, t=t, sub_frac=sub_frac, top_frac=top_frac,
                               mask=mask, pools=pools, p=p, seed=seed, gen=self._gen)
        self.last_interest = (u1, u2, ug)
        return scores

    def backward(self, flat, gflat, dscores, gen=None, table_grad_ready=None):
        sv = self._saved
        if sv is None:
            raise _lib.NrmsError("backward() without a training forward()")
        if gen is not None and gen != sv["gen"]:
            raise _lib.NrmsError("backward() for training forward #%d, but the saved activations belong to forward #%d" % (gen, sv["gen"]))
        B, H, Cn, L = sv["B"], sv["H"], sv["C"], sv["L"]
        d = self.dims.word_embed_size
        n, N = B * H, B * (H + Cn)
        t, lay = sv["t"], self.layout
        self.poll_grad_overflow()
        self.loss_scale = float(getattr(self, "loss_scale_override", None) or -float(self.loss_scale_backoff))
        dnv = self._buf("d_news_vec", N * d)[:N * d].view(N, d)
        du1 = self._buf("d_u1", n * d)[:n * d].view(n, d)
        du2 = self._buf("d_u2", n * d)[:n * d].view(n, d)
        dug = self._buf("d_ug", B * d)[:B * d].view(B, d)
        du1.zero_()
        du2.zero_()
        cand = sv["nv"][n:]
        rc = self.lib.nrms_hier_score_bwd(B, Cn, d, _lib.ptr(cand), _lib.ptr(sv["u1"]), _lib.ptr(sv["u2"]), _lib.ptr(sv["ug"]),
                                          _lib.ptr(t["sub_slot"]), _lib.ptr(sv["sub_frac"]), _lib.ptr(t["top_slot"]), _lib.ptr(sv["top_frac"]),
                                          _lib.ptr(sv["mask"]), C.c_float(self.lambda_sub), C.c_float(self.lambda_top),
                                          _lib.ptr(dscores.contiguous()), C.c_void_p(dnv[n:].data_ptr()), _lib.ptr(du1), _lib.ptr(du2),
                                          _lib.ptr(dug), _stream())
        _lib.check(rc, "nrms_hier_score_bwd")
        g = lambda name: lay.view(gflat, name)
        f = lambda name: lay.view(flat, name)
        pools = sv["pools"]

        def level(i, dout):
            lv = LEVELS[i]
            return pools[i].backward(f(lv + ".linear.weight"), f(lv + ".attention_query_vector"), dout, g(lv + ".linear.weight"),
                                     g(lv + ".linear.bias"), g(lv + ".attention_query_vector"))

        eb = max(int(self.lib.nrms_hier_add_embedding_bwd_workspace_bytes(C.c_int64(n), d, k)) for k in (self.n_top, self.n_sub))
        ews = self._buf("hier_embgrad_ws", (eb + 3) // 4 + 1)
        du2 += level(2, dug)                                   # user level -> topic interests
        _lib.check(self.lib.nrms_hier_add_embedding_bwd(C.c_int64(n), d, self.n_top, _lib.ptr(t["l2_top"]), _lib.ptr(t["l2_cnt"]), _lib.ptr(du2),
                                                        _lib.ptr(g("topic_embedding.weight")), _lib.ptr(ews), C.c_size_t(ews.numel() * 4),
                                                        _stream()), "add_embedding_bwd")
        du1 += level(1, du2)                                   # topic level -> sub-topic interests
        _lib.check(self.lib.nrms_hier_add_embedding_bwd(C.c_int64(n), d, self.n_sub, _lib.ptr(t["l1_sub"]), _lib.ptr(t["l1_cnt"]), _lib.ptr(du1),
                                                        _lib.ptr(g("subtopic_embedding.weight")), _lib.ptr(ews), C.c_size_t(ews.numel() * 4),
                                                        _stream()), "add_embedding_bwd")
        dnv[:n].copy_(level(0, du1))                           # sub-topic level -> news vectors of the history
        # ---- news encoder (the NRMS engine's own backward, model/nrms_v0.py:154-176)
        desc_n = self._desc("news_encoder", N, L, sv["p"], sv["p"], sv["seed"], training=True)
        ws = self._bwd_workspace(desc_n)
        wn, gn = self._weights(flat, "news_encoder"), self._grads(gflat, "news_encoder")
        acts_n = self._acts("news", N * L, True, gather=True, desc=desc_n)
        if desc_n.precision == _lib.NRMS_PRECISION_FP16:
            desc_n.flags |= _lib.NRMS_FLAG_FWD_SCRATCH_KEPT
        rc = self.lib.nrms_encoder_bwd(C.byref(desc_n), C.byref(wn), _lib.ptr(sv["ids"]), None, None, C.byref(acts_n), _lib.ptr(dnv),
                                       C.byref(gn), None, _lib.ptr(ws), C.c_size_t(ws.numel() * 4), _stream())
        _lib.check(rc, "nrms_encoder_bwd(news)")
        if table_grad_ready is not None:
            table_grad_ready()


class Model(FlatHipModel):
    """HieRec-style hierarchical interest model: ``Model(config)``, ``forward(batch) -> scores [B, C]`` (the reference's plugin
    contract, ``model/__init__.py:22-23,38``)."""
    KEYS = ("browsed_titles", "browsed_categ_ids", "browsed_subcateg_ids", "browsed_mask", "candidate_titles", "candidate_categ_ids",
            "candidate_subcateg_ids", "candidate_mask")

    def __init__(self, config, pretrained_word_embedding=None):
        super().__init__()
        self.config = config
        table = nrms_hip._load_table(config, pretrained_word_embedding)
        V, d = table.shape
        q = int(config.query_vector_dim)
        self.news_encoder = nrms_hip._NewsEncoderParams(config, table)
        self.subtopic_embedding = nn.Embedding(int(config.subcategory_nums), d, padding_idx=0)
        self.topic_embedding = nn.Embedding(int(config.category_nums), d, padding_idx=0)
        self.subtopic_attention = AdditiveParams(q, d)
        self.topic_attention = AdditiveParams(q, d)
        self.user_attention = AdditiveParams(q, d)
        self._dims = ModelDims(n_words=int(V), word_embed_size=int(d), num_attention_heads=int(config.num_attention_heads), query_vector_dim=q)
        self._finish(hier_layout(self._dims, int(config.subcategory_nums), int(config.category_nums)), table.device)

    def _make_engine(self, device, precision):
        return HieRecEngine(self._dims, self._layout, device, precision, self.config.subcategory_nums, self.config.category_nums,
                            getattr(self.config, "hierec_lambda_sub", 0.7), getattr(self.config, "hierec_lambda_top", 0.15))

    def _zero_frozen_rows(self, gflat):
        # padding_idx = 0 of the two embedding tables: row 0 takes no gradient (nn.Embedding semantics)
        self._layout.view(gflat, "subtopic_embedding.weight")[0].zero_()
        self._layout.view(gflat, "topic_embedding.weight")[0].zero_()
