"""User-news graph encoder on MI355X (BASELINE configs[4]): neighbour gather + attention aggregate.

PARITY UNPINNED.  The reference repository holds no graph model (``/root/reference/README.md:3`` names Adressa, no code follows;
``model/tanr.py`` is empty), so there is nothing of the reference's to match; the model is specified here, as a two-layer
attention aggregation over a sampled sub-graph of the bipartite click graph (the aggregation of GERL, Ge et al., WWW 2020, with
the reference's additive attention, ``model/nrms_v0.py:100-126``, as the aggregator), and checked against
``oracle/segpool_oracle.py``.  It keeps the reference's plugin API -- ``Model(config)``, ``forward(batch_dict) -> [B, C]`` with
masked candidates at -1e9 -- and reads, beside the NRMS keys of ``data_handler.py:236-250`` (``browsed_titles``,
``browsed_mask``, ``candidate_titles``, ``candidate_mask``), one new key:

  ``neighbor_rows`` [B * (H + C), K] int64 -- for every news slot of the batch (row r < B * H: history slot (r // H, r % H); row
  B * H + b * C + c: candidate c of user b) up to K sampled neighbour news (news clicked by the users who clicked it), as ROWS of
  the same numbering; -1 = none.  The sampler draws neighbours from the batch's own news (an induced sub-graph), so a
  data-parallel rank needs no remote rows: users -- and their sub-graphs -- shard across GPUs, gradients all-reduce (RCCL).

Specification (n_r = NRMS news encoder of slot r's title, ``nrms_v0.py:154-176``; AddPool as in model/hierec_hip.py):
  news layer:  g_r = n_r + AddPool({n_k : k in neighbor_rows[r]}; neighbor_attention)          (no neighbours: g_r = n_r)
  user layer:  h_b = AddPool({g_r : r a history slot of b with browsed_mask = 1}; user_attention)  (no click: 0)
  score(b, c) = <g_cand(b, c), h_b>                                                               (``nrms_v0.py:205-216``)
Both aggregations are ``nrms_segment_pool_fwd / _bwd`` (csrc/segpool.hip) over index lists built on the device by
``nrms_csr_from_padded``; a news row is listed by many segments, so the news layer's backward sorts the list entries by row and
adds a row's shares in list order (no atomics: bit-reproducible, include/nrms_hip.h).  A batch without ``neighbor_rows`` (the
reference's loader knows no graph) gets them from ``graph_sampler.induced_neighbor_rows``.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from ..engine import ModelDims, NRMSEngine, _stream
from ..segpool import SegmentPool
from . import nrms_hip
from ._flat_model import AdditiveParams, FlatHipModel, FlatLayout2, additive_entries

LEVELS = ("neighbor_attention", "user_attention")


class GraphEngine(NRMSEngine):
    """Forward / backward of the graph encoder on one GPU: the NRMS engine's news encoder, two segment pools, click scores."""

    def __init__(self, dims, layout, device, precision):
        super().__init__(dims, device, precision=precision)
        self.layout = layout

    def _csr(self, key, lists, n_rows):
        n_seg, K = lists.shape
        ptr = self._buf(key + "_ptr", n_seg + 1, torch.int32)[:n_seg + 1]
        idx = self._buf(key + "_idx", n_seg * K, torch.int32)[:n_seg * K]
        rc = self.lib.nrms_csr_from_padded(C.c_int64(n_seg), int(K), _lib.ptr(lists), C.c_int64(n_rows), _lib.ptr(ptr), _lib.ptr(idx), _stream())
        _lib.check(rc, "nrms_csr_from_padded")
        return ptr, idx

    def forward(self, flat, batch, training, p_drop=0.0, seed=0):
        bt, ct = batch["browsed_titles"], batch["candidate_titles"]
        B, H, L = bt.shape
        Cn = ct.shape[1]
        d, q = self.dims.word_embed_size, self.dims.query_vector_dim
        N = B * (H + Cn)
        nbr = batch["neighbor_rows"].to(torch.int64).contiguous()
        if nbr.dim() != 2 or nbr.shape[0] != N:
            raise _lib.NrmsError("graph: neighbor_rows must be [B * (H + C) = %d, K], got %s" % (N, tuple(nbr.shape)))
        sfx = "" if training else "_eval"
        self.poll_ids()
        ids = self._buf("ids" + sfx, N * L, torch.int64)[:N * L].view(N, L)
        self.sanitize_ids(bt.reshape(B * H, L).contiguous(), ids[:B * H])
        self.sanitize_ids(ct.reshape(B * Cn, L).contiguous(), ids[B * H:])
        nv = self._buf("news_vec" + sfx, N * d)[:N * d].view(N, d)
        p = float(p_drop)
        self.encode_titles(flat, ids, out=nv, p_embed=p, p_ctx=p, seed=seed, save=training, tag="news" + sfx, trusted_ids=True)
        # ---- index lists: every slot's neighbours; every user's clicked slots (history rows with browsed_mask = 1)
        n_ptr, n_idx = self._csr("nbr" + sfx, nbr, N)
        slots = torch.arange(B * H, device=self.device, dtype=torch.int64).view(B, H)
        hist = torch.where(batch["browsed_mask"].to(torch.bool), slots, torch.full_like(slots, -1)).contiguous()
        u_ptr, u_idx = self._csr("hist" + sfx, hist, B * H)
        lay = self.layout
        W = {lv: (lay.view(flat, lv + ".linear.weight"), lay.view(flat, lv + ".linear.bias"), lay.view(flat, lv + ".attention_query_vector"))
             for lv in LEVELS}
        prec = "fp32" if self.precision == "fp32" else "bf16x3"
        pool_n = SegmentPool(d, q, prec, rows_unique=False)
        pool_u = SegmentPool(d, q, prec, rows_unique=True)
        g = pool_n.forward(nv, *W[LEVELS[0]], n_ptr, n_idx)          # [N, d]: the neighbour aggregate ...
        g += nv                                                      # ... + the slot's own vector
        h = pool_u.forward(g, *W[LEVELS[1]], u_ptr, u_idx)           # [B, d]
        mask = batch.get("candidate_mask")
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        scores = torch.empty(B, Cn, dtype=torch.float32, device=self.device)
        self.click_scores(g[B * H:].view(B, Cn, d), h, mask, out=scores)
        self._bad_host.copy_(self._bad_ids, non_blocking=True)
        if training:
            self._gen += 1
            self._saved = dict(B=B, H=H, C=Cn, L=L, ids=ids, g=g, h=h, mask=mask, pools=(pool_n, pool_u), p=p, seed=seed, gen=self._gen)
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
        lay = self.layout
        self.poll_grad_overflow()
        self.loss_scale = float(getattr(self, "loss_scale_override", None) or -float(self.loss_scale_backoff))
        g, h = sv["g"], sv["h"]
        dcand = self._buf("d_cand", B * Cn * d)[:B * Cn * d].view(B * Cn, d)
        dh = self._buf("d_user_vec", B * d)[:B * d].view(B, d)
        rc = self.lib.nrms_click_score_bwd(B, Cn, d, C.c_void_p(g[n:].data_ptr()), _lib.ptr(h), _lib.ptr(sv["mask"]), _lib.ptr(dscores.contiguous()),
                                           _lib.ptr(dcand), _lib.ptr(dh), _stream())
        _lib.check(rc, "nrms_click_score_bwd")
        gv = lambda name: lay.view(gflat, name)
        fv = lambda name: lay.view(flat, name)
        pool_n, pool_u = sv["pools"]

        def level(pool, lv, dout):
            return pool.backward(fv(lv + ".linear.weight"), fv(lv + ".attention_query_vector"), dout, gv(lv + ".linear.weight"),
                                 gv(lv + ".linear.bias"), gv(lv + ".attention_query_vector"))

        dg = level(pool_u, LEVELS[1], dh)                            # [N, d] (rows outside every history list: 0)
        dg[n:] += dcand
        dnv = level(pool_n, LEVELS[0], dg)                           # through the neighbour aggregate ...
        dnv += dg                                                    # ... and the slot's own vector
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
    """User-news graph encoder: ``Model(config)``, ``forward(batch) -> scores [B, C]`` (``model/__init__.py:22-23,38``)."""
    KEYS = ("browsed_titles", "browsed_mask", "candidate_titles", "candidate_mask", "neighbor_rows")

    def __init__(self, config, pretrained_word_embedding=None):
        super().__init__()
        self.config = config
        table = nrms_hip._load_table(config, pretrained_word_embedding)
        V, d = table.shape
        q = int(config.query_vector_dim)
        self.news_encoder = nrms_hip._NewsEncoderParams(config, table)
        self.neighbor_attention = AdditiveParams(q, d)
        self.user_attention = AdditiveParams(q, d)
        self._dims = ModelDims(n_words=int(V), word_embed_size=int(d), num_attention_heads=int(config.num_attention_heads), query_vector_dim=q)
        extra = []
        for lv in LEVELS:
            extra += additive_entries(lv, q, d)
        self._finish(FlatLayout2(self._dims, extra), table.device)

    def _make_engine(self, device, precision):
        return GraphEngine(self._dims, self._layout, device, precision)

    def _device_batch(self, batch, dev):
        if (batch.get("neighbor_rows") if hasattr(batch, "get") else None) is None:
            # a loader that knows no graph (data_handler.MyDataset): sample the neighbours from the click graph induced on this batch
            from ..graph_sampler import induced_neighbor_rows
            self._sampled = getattr(self, "_sampled", 0) + 1
            cpu = lambda v: torch.as_tensor(v).cpu().numpy()
            batch = dict(batch)
            batch["neighbor_rows"] = induced_neighbor_rows(cpu(batch["browsed_titles"]), cpu(batch["browsed_mask"]), cpu(batch["candidate_titles"]),
                                                           int(getattr(self.config, "graph_neighbors", 8)), seed=self._sampled)
        return super()._device_batch(batch, dev)
