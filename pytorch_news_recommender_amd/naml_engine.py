"""Host-side driver of the nrms_naml variant (/root/reference/MIND_2020/model/nrms_naml.py) over the
same C ABI (include/nrms_hip.h): two passes of the word-level encoder (title, abstract: shared weights, W_O, dropout on
the attention probabilities), the news feature rows with the category embeddings, LayerNorm on the history, the wide
user encoder, click scores.

Layout in HBM (fp32):
  flat parameter / gradient buffer
      [ word table V*d | category table | sub-category table
      | news: Wq|Wk|Wv (3d*d), bq|bk|bv, Wo, bo, Wa (q*d), ba, qv | user: same with F = 2d + 2c and Q | norm.weight | norm.bias ]
  slots   N = B*H history slots (user-major) then B*C candidate slots, as in the NRMS engine
  feat    [N, F]: rows [0, B*H) feed LayerNorm -> user encoder, rows [B*H, N) are the candidate vectors
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from .engine import NRMSEngine, _stream

ENCODERS = ("news_encoder", "user_encoder")


@dataclass(frozen=True)
class NamlDims:
    """config.py:45-49,68-72,77,87 as nrms_naml.py reads them."""
    n_words: int
    word_embed_size: int
    title_heads_num: int
    query_vector_dim: int
    category_nums: int
    subcategory_nums: int
    cate_embed_size: int
    user_heads_num: int
    query_vector_dim_large: int
    style: str = "naml"

    @property
    def news_feature_size(self):
        return 2 * self.word_embed_size + 2 * self.cate_embed_size

    def width(self, enc):
        return self.word_embed_size if enc == "news_encoder" else self.news_feature_size

    def heads(self, enc):
        return self.title_heads_num if enc == "news_encoder" else self.user_heads_num

    def q(self, enc):
        return self.query_vector_dim if enc == "news_encoder" else self.query_vector_dim_large


class NamlLayout:
    """Offsets (floats) of the reference-named tensors of nrms_naml.Model.state_dict() inside the flat buffer."""

    def __init__(self, dims: NamlDims):
        self.dims = dims
        d, c = dims.word_embed_size, dims.cate_embed_size
        if d % 4 or c % 4 or dims.query_vector_dim % 4 or dims.query_vector_dim_large % 4:
            raise ValueError("word_embed_size, cate_embed_size and the query vector sizes must be multiples of 4")
        self.entries, self.blocks = {}, {enc: {} for enc in ENCODERS}
        off = 0

        def put(name, shape, enc=None, role=None):
            nonlocal off
            n = 1
            for x in shape:
                n *= x
            self.entries[name] = (off, tuple(shape), n)
            if enc is not None:
                self.blocks[enc][role] = off
            off += n

        put("news_encoder.word_embedding.weight", (dims.n_words, d), "news_encoder", "table")
        put("news_encoder.category_embedding.weight", (dims.category_nums, c))
        put("news_encoder.subcategory_embedding.weight", (dims.subcategory_nums, c))
        for enc in ENCODERS:
            w, q = dims.width(enc), dims.q(enc)
            a = enc + ".multi_head_self_attention."
            for i, r in enumerate(("wq", "wk", "wv")):
                put(a + "linear_layers.%d.weight" % i, (w, w), enc, r)
            for i, r in enumerate(("bq", "bk", "bv")):
                put(a + "linear_layers.%d.bias" % i, (w,), enc, r)
            put(a + "output_linear.weight", (w, w), enc, "wo")
            put(a + "output_linear.bias", (w,), enc, "bo")
            put(enc + ".additive_attention.linear.weight", (q, w), enc, "wa")
            put(enc + ".additive_attention.linear.bias", (q,), enc, "ba")
            put(enc + ".additive_attention.query_vector", (q,), enc, "qv")
        put("norm.weight", (dims.news_feature_size,))
        put("norm.bias", (dims.news_feature_size,))
        assert self.entries["norm.bias"][0] == self.entries["norm.weight"][0] + dims.news_feature_size
        self.total = off
        self.names = list(self.entries)
        self.table = 0

    def view(self, flat, name):
        off, shp, n = self.entries[name]
        return flat[off:off + n].view(shp)


class NamlEngine(NRMSEngine):
    """One nrms_naml forward / backward on one GPU.  Inherits the shape-independent pieces of the NRMS engine (buffers,
    id validation, click scores, CE, Adam, AUC, timers)."""

    def __init__(self, dims: NamlDims, device, precision="fp32"):
        self.lib = _lib.load()
        self.dims = dims
        self.layout = NamlLayout(dims)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NrmsError("the NRMS HIP engine needs a GPU device (got %s); there is no CPU path" % device)
        self.set_precision(precision)
        self._bufs = {}
        self._saved = None
        self._gen = 0
        self.loss_scale = 0.0
        self._bad_ids = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._bad_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._bad_event = None
        self._news_cache = None
        # all-padding title / abstract sequences (history padding slots: 41 % of a MIND-shaped batch) in closed form
        # (nrms_encoder_empty_fwd / _bwd) while the kernel chain runs on the others, compacted; needs a zero padding row
        self.closed_form_empty = True

    def set_precision(self, precision):
        """fp32 or bf16x3 projections; the fused fp16 kernels have no W_O / attention-probability dropout, so "fp16"
        runs this variant in bf16x3."""
        if precision not in _lib.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        self.precision = "bf16x3" if precision == "fp16" else precision
        self.pad_row_zero = False

    # ---- descriptors and pointers --------------------------------------------------------------------------------
    def _desc(self, enc, n_seq, seq_len, p_attn=0.0, seed=0):
        d = self.dims
        return _lib.EncoderDesc(n_seq=n_seq, seq_len=seq_len, d_model=d.width(enc), n_heads=d.heads(enc), q_dim=d.q(enc),
                                vocab=d.n_words if enc == "news_encoder" else 0, p_drop_embed=0.0, p_drop_ctx=0.0,
                                precision=_lib.PRECISIONS[self.precision], use_output_proj=1, mask_mode=0,
                                flags=(_lib.NRMS_FLAG_PAD_ROW_ZERO if (self.pad_row_zero and enc == "news_encoder") else 0),
                                seed=int(seed) & 0xFFFFFFFFFFFFFFFF, loss_scale=0.0, p_drop_attn=float(p_attn))

    def _ptrs(self, cls, flat, enc):
        b = self.layout.blocks[enc]
        base = flat.data_ptr()
        p = lambda role: (base + 4 * b[role]) if role in b else None
        return cls(table=p("table"), w_qkv=p("wq"), b_qkv=p("bq"), w_o=p("wo"), b_o=p("bo"), w_add=p("wa"),
                   b_add=p("ba"), q_vec=p("qv"))

    def _acts(self, tag, desc, gather):
        M, d, q = desc.n_seq * desc.seq_len, desc.d_model, desc.q_dim
        dp = lambda z: None if z is None else z.data_ptr()
        x = self._buf(tag + ".x", M * d) if gather else None
        nbytes = int(self.lib.nrms_encoder_fwd_scratch_bytes(C.byref(desc)))
        return _lib.EncoderActs(x=dp(x), qkv=dp(self._buf(tag + ".qkv", M * 3 * d)), attn=dp(self._buf(tag + ".attn", M * d)),
                                ctx=dp(self._buf(tag + ".ctx", M * d)), t=dp(self._buf(tag + ".t", M * q)),
                                w=dp(self._buf(tag + ".w", M)), scratch=dp(self._buf("fwd_scratch", (nbytes + 3) // 4)))

    def _off(self, name):
        return self.layout.entries[name][0]

    # ---- pieces --------------------------------------------------------------------------------------------------
    def _split_ok(self, L):
        d = self.dims
        return (self.closed_form_empty and self.pad_row_zero and L <= 64 and d.heads("news_encoder") <= 8 and d.word_embed_size <= 512
                and d.q("news_encoder") <= 256 and d.word_embed_size // d.heads("news_encoder") <= 64
                and (d.word_embed_size // d.heads("news_encoder")) % 2 == 0)

    def partition(self, ids, tag):
        """Device lists of the sequences with a real token / the all-padding ones (nrms_sequence_partition): (order, counts)."""
        N, L = ids.shape
        order = self._buf(tag + ".order", 2 * N, torch.int32)[:2 * N]
        counts = self._buf(tag + ".order_cnt", int(self.lib.nrms_sequence_partition_count_ints(N)) + 2, torch.int32)
        rc = self.lib.nrms_sequence_partition(_lib.ptr(ids), N, L, _lib.ptr(order), _lib.ptr(counts), _stream())
        _lib.check(rc, "nrms_sequence_partition")
        return order, counts

    def encode_text(self, flat, ids, tag, p_attn=0.0, seed=0, out=None, split=None):
        """ids [N, L] (validated) -> [N, d]: embedding, MHSA with W_O, additive attention (nrms_naml.py:152-158).
        split = (order, n_real) from partition(): the kernel chain runs on the n_real sequences with a real token (gathered; their
        dropout counters stay those of the full batch, desc.seq_index), the all-padding ones take the closed form."""
        N, L = ids.shape
        d = self.dims.word_embed_size
        if out is None:
            out = torch.empty(N, d, dtype=torch.float32, device=self.device)
        w = self._ptrs(_lib.EncoderWeights, flat, "news_encoder")
        if split is None:
            desc = self._desc("news_encoder", N, L, p_attn, seed)
            acts = self._acts(tag, desc, gather=True)
            rc = self.lib.nrms_encoder_fwd(C.byref(desc), C.byref(w), _lib.ptr(ids), None, None, C.byref(acts), _lib.ptr(out), _stream())
            _lib.check(rc, "nrms_encoder_fwd(%s)" % tag)
            return out
        order, n_real = split
        n_empty = N - n_real
        if n_real:
            rows = order[:n_real].to(torch.int64)
            ids_c = self._buf(tag + ".ids_c", N * L, torch.int64)[:n_real * L].view(n_real, L)
            torch.index_select(ids, 0, rows, out=ids_c)
            desc = self._desc("news_encoder", n_real, L, p_attn, seed)
            desc.seq_index = order.data_ptr()
            acts = self._acts(tag, desc, gather=True)
            out_c = self._buf(tag + ".out_c", N * d)[:n_real * d].view(n_real, d)
            rc = self.lib.nrms_encoder_fwd(C.byref(desc), C.byref(w), _lib.ptr(ids_c), None, None, C.byref(acts), _lib.ptr(out_c), _stream())
            _lib.check(rc, "nrms_encoder_fwd(%s)" % tag)
            out.index_copy_(0, rows, out_c)
        if n_empty:
            elist = order[N:N + n_empty]
            desc_e = self._desc("news_encoder", n_empty, L, p_attn, seed)
            ews = self._empty_ws(desc_e)
            out_e = self._buf(tag + ".out_e", N * d)[:n_empty * d].view(n_empty, d)
            saved = self._buf(tag + ".empty_saved", (int(self.lib.nrms_encoder_empty_saved_bytes(C.byref(desc_e))) + 3) // 4)
            rc = self.lib.nrms_encoder_empty_fwd(C.byref(desc_e), C.byref(w), _lib.ptr(elist), _lib.ptr(out_e), _lib.ptr(saved), _lib.ptr(ews),
                                                 C.c_size_t(ews.numel() * 4), _stream())
            _lib.check(rc, "nrms_encoder_empty_fwd(%s)" % tag)
            out.index_copy_(0, elist.to(torch.int64), out_e)
        return out

    def _empty_ws(self, desc_e):
        nb = int(self.lib.nrms_encoder_empty_workspace_bytes(C.byref(desc_e)))
        if nb == 0:
            _lib.check(-1, "nrms_encoder_empty_workspace_bytes")
        return self._buf("empty_ws", (nb + 3) // 4 + 64)

    def encode_text_backward(self, flat, gflat, ids, tag, dout, p_attn, seed, ws, split=None):
        """Backward of encode_text (training forward): accumulates the word-level encoder's gradients into gflat."""
        N, L = ids.shape
        wn, gn = self._ptrs(_lib.EncoderWeights, flat, "news_encoder"), self._ptrs(_lib.EncoderGrads, gflat, "news_encoder")
        if split is None:
            desc = self._desc("news_encoder", N, L, p_attn, seed)
            rc = self.lib.nrms_encoder_bwd(C.byref(desc), C.byref(wn), _lib.ptr(ids), None, None, C.byref(self._acts(tag, desc, gather=True)),
                                           _lib.ptr(dout), C.byref(gn), None, _lib.ptr(ws), C.c_size_t(ws.numel() * 4), _stream())
            _lib.check(rc, "nrms_encoder_bwd(%s)" % tag)
            return
        order, n_real = split
        n_empty = N - n_real
        d = self.dims.word_embed_size
        if n_real:
            rows = order[:n_real].to(torch.int64)
            ids_c = self._buf(tag + ".ids_c", N * L, torch.int64)[:n_real * L].view(n_real, L)      # (as the forward left it)
            dout_c = self._buf(tag + ".dout_c", N * d)[:n_real * d].view(n_real, d)
            torch.index_select(dout, 0, rows, out=dout_c)
            desc = self._desc("news_encoder", n_real, L, p_attn, seed)
            desc.seq_index = order.data_ptr()
            rc = self.lib.nrms_encoder_bwd(C.byref(desc), C.byref(wn), _lib.ptr(ids_c), None, None, C.byref(self._acts(tag, desc, gather=True)),
                                           _lib.ptr(dout_c), C.byref(gn), None, _lib.ptr(ws), C.c_size_t(ws.numel() * 4), _stream())
            _lib.check(rc, "nrms_encoder_bwd(%s)" % tag)
        if n_empty:
            elist = order[N:N + n_empty]
            dout_e = self._buf(tag + ".dout_e", N * d)[:n_empty * d].view(n_empty, d)
            torch.index_select(dout, 0, elist.to(torch.int64), out=dout_e)
            desc_e = self._desc("news_encoder", n_empty, L, p_attn, seed)
            ews = self._empty_ws(desc_e)
            saved = self._buf(tag + ".empty_saved", (int(self.lib.nrms_encoder_empty_saved_bytes(C.byref(desc_e))) + 3) // 4)   # (as the forward left it)
            rc = self.lib.nrms_encoder_empty_bwd(C.byref(desc_e), C.byref(wn), _lib.ptr(elist), _lib.ptr(dout_e), _lib.ptr(saved), C.byref(gn),
                                                 _lib.ptr(ews), C.c_size_t(ews.numel() * 4), _stream())
            _lib.check(rc, "nrms_encoder_empty_bwd(%s)" % tag)

    def _features_desc(self, flat, n, tv, av, categ, subcateg, p_drop, seed):
        d = self.dims
        base = flat.data_ptr()
        return _lib.NewsFeatures(n=n, d_text=d.word_embed_size, d_cat=d.cate_embed_size, n_cat=d.category_nums,
                                 n_sub=d.subcategory_nums, p_drop=float(p_drop), seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
                                 title_vec=_lib.ptr(tv).value if tv is not None else None,
                                 abst_vec=_lib.ptr(av).value if av is not None else None,
                                 cat_table=base + 4 * self._off("news_encoder.category_embedding.weight"),
                                 sub_table=base + 4 * self._off("news_encoder.subcategory_embedding.weight"),
                                 categ=categ.data_ptr(), subcateg=subcateg.data_ptr())

    def news_features(self, flat, ids_t, ids_a, categ, subcateg, p_drop=0.0, seed=0, sfx="", out=None):
        """NewsEncoder.forward (nrms_naml.py:121-177) for N slots at once -> [N, F]."""
        N = ids_t.shape[0]
        d = self.dims
        tv = self._buf("title_vec" + sfx, N * d.word_embed_size)[:N * d.word_embed_size].view(N, -1)
        av = self._buf("abst_vec" + sfx, N * d.word_embed_size)[:N * d.word_embed_size].view(N, -1)
        split_t = split_a = None
        if self._split_ok(ids_t.shape[1]) and self._split_ok(ids_a.shape[1]):
            # the lists of both passes, then ONE read of their sizes (the only host synchronisation of the step: the chain's grids
            # are sized by the number of sequences that hold a real token)
            (ot, ct), (oa, ca) = self.partition(ids_t, "title" + sfx), self.partition(ids_a, "abst" + sfx)
            n_t, n_a = (int(v) for v in torch.stack([ct[0], ca[0]]).cpu())
            split_t, split_a = (ot, n_t), (oa, n_a)
        self.last_split = (split_t, split_a)
        self.encode_text(flat, ids_t, "title" + sfx, p_drop, seed, out=tv, split=split_t)
        self.encode_text(flat, ids_a, "abst" + sfx, p_drop, seed ^ 0x5DEECE66D1CE4E5B, out=av, split=split_a)
        if out is None:
            out = torch.empty(N, d.news_feature_size, dtype=torch.float32, device=self.device)
        f = self._features_desc(flat, N, tv, av, categ, subcateg, p_drop, seed)
        _lib.check(self.lib.nrms_news_features_fwd(C.byref(f), _lib.ptr(out), _stream()), "nrms_news_features_fwd")
        return out

    def layernorm(self, flat, x, stats=None, out=None, eps=1e-5):
        n, F = x.shape
        if out is None:
            out = torch.empty_like(x)
        base = flat.data_ptr()
        rc = self.lib.nrms_layernorm_fwd(C.c_int64(n), F, _lib.ptr(x), C.c_void_p(base + 4 * self._off("norm.weight")),
                                         C.c_void_p(base + 4 * self._off("norm.bias")), C.c_float(eps), _lib.ptr(out),
                                         _lib.ptr(stats), _stream())
        _lib.check(rc, "nrms_layernorm_fwd")
        return out

    def encode_users(self, flat, x, p_attn=0.0, seed=0, tag="user", out=None):
        """x [B, H, F] (normalised history) -> [B, F] (UserEncoder.forward, nrms_naml.py:188-191)."""
        B, H, F = x.shape
        if out is None:
            out = torch.empty(B, F, dtype=torch.float32, device=self.device)
        desc = self._desc("user_encoder", B, H, p_attn, seed)
        acts = self._acts(tag, desc, gather=False)
        w = self._ptrs(_lib.EncoderWeights, flat, "user_encoder")
        rc = self.lib.nrms_encoder_fwd(C.byref(desc), C.byref(w), None, _lib.ptr(x), None, C.byref(acts), _lib.ptr(out), _stream())
        _lib.check(rc, "nrms_encoder_fwd(user)")
        return out

    def _slot_inputs(self, batch, sfx):
        """History slots then candidate slots: validated copies of the four id tensors."""
        d = self.dims
        bt, ct = batch["browsed_titles"], batch["candidate_titles"]
        B, H, Lt = bt.shape
        Cn = ct.shape[1]
        La = batch["browsed_absts"].shape[2]
        N = B * (H + Cn)
        self.poll_ids()

        def both(kb, kc, width, name, vocab):
            dst = self._buf(name + sfx, N * width, torch.int64)[:N * width].view(N, width)
            for src, lo, hi in ((batch[kb], 0, B * H), (batch[kc], B * H, N)):
                src = src.reshape(hi - lo, width).contiguous()
                if hi > lo:
                    rc = self.lib.nrms_sanitize_ids(_lib.ptr(src), _lib.ptr(dst[lo:hi]), C.c_int64(src.numel()), int(vocab),
                                                    _lib.ptr(self._bad_ids), _stream())
                    _lib.check(rc, "nrms_sanitize_ids")
            return dst

        ids_t = both("browsed_titles", "candidate_titles", Lt, "ids_title", d.n_words)
        ids_a = both("browsed_absts", "candidate_absts", La, "ids_abst", d.n_words)
        categ = both("browsed_categ_ids", "candidate_categ_ids", 1, "ids_categ", d.category_nums).view(N)
        subcateg = both("browsed_subcateg_ids", "candidate_subcateg_ids", 1, "ids_subcateg", d.subcategory_nums).view(N)
        self._bad_host.copy_(self._bad_ids, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._bad_event = ev
        return (B, H, Cn, N), ids_t, ids_a, categ, subcateg

    def _raise_bad_ids(self):
        n = int(self._bad_host.item())
        if n:
            self._bad_ids.zero_()
            self._bad_host.zero_()
            raise _lib.NrmsError("%d word / category id(s) outside their table reached the embedding gather (treated as "
                                 "the padding id on the device)" % n)

    # ---- full model ------------------------------------------------------------------------------------------------
    def forward(self, flat, batch, cand_mask, training, p_drop=0.0, seed=0):
        """Model.forward (nrms_naml.py:216-257).  batch: dict of int64 device tensors (the eight id tensors);
        cand_mask [B, C] uint8 or None.  -> scores [B, C]."""
        sfx = "" if training else "_eval"
        (B, H, Cn, N), ids_t, ids_a, categ, subcateg = self._slot_inputs(batch, sfx)
        F = self.dims.news_feature_size
        if not training and p_drop == 0.0 and getattr(self, "dedup_inference", True):
            return self._forward_dedup(flat, (B, H, Cn, N), ids_t, ids_a, categ, subcateg, cand_mask)
        feat = self._buf("feat" + sfx, N * F)[:N * F].view(N, F)
        self.news_features(flat, ids_t, ids_a, categ, subcateg, p_drop, seed, sfx, out=feat)
        stats = self._buf("ln_stats" + sfx, 2 * B * H)[:2 * B * H]
        normed = self._buf("normed" + sfx, B * H * F)[:B * H * F].view(B * H, F)
        self.layernorm(flat, feat[:B * H], stats, out=normed)
        user = self._buf("user_vec" + sfx, B * F)[:B * F].view(B, F)
        self.encode_users(flat, normed.view(B, H, F), p_drop, seed ^ 0x2545F4914F6CDD1D, "user" + sfx, out=user)
        if cand_mask is not None:
            cand_mask = cand_mask.contiguous()
        scores = self.click_scores(feat[B * H:].view(B, Cn, F), user, cand_mask)
        if training:
            self._gen += 1
            self._saved = dict(B=B, H=H, C=Cn, N=N, ids_t=ids_t, ids_a=ids_a, categ=categ, subcateg=subcateg, feat=feat,
                               stats=stats, normed=normed, user=user, mask=cand_mask, p=float(p_drop), seed=seed, gen=self._gen,
                               split=self.last_split)
        return scores

    def _forward_dedup(self, flat, dims, ids_t, ids_a, categ, subcateg, cand_mask):
        """Inference over the DISTINCT news of the batch (an evaluation impression is padded to 300 candidate slots,
        data_handler.py:174-177, mostly padding and repeats): slots are grouped exactly by their (title, abstract, category,
        sub-category) ids on the device (nrms_title_dedup), one feature row is computed per group, the history rows are
        gathered for LayerNorm and the user encoder, the candidates are scored by index."""
        B, H, Cn, N = dims
        F = self.dims.news_feature_size
        key = torch.cat([ids_t, ids_a, categ.view(N, 1), subcateg.view(N, 1)], 1).contiguous()
        inverse, rep, U = self.group_rows(key)
        r = rep.to(torch.int64)
        feat = self.news_features(flat, ids_t.index_select(0, r), ids_a.index_select(0, r), categ.index_select(0, r),
                                  subcateg.index_select(0, r), 0.0, 0, "_uniq")
        self.last_unique_news = U
        hist = feat.index_select(0, inverse[:B * H])
        normed = self.layernorm(flat, hist)
        user = self.encode_users(flat, normed.view(B, H, F), 0.0, 0, "user_eval")
        if cand_mask is not None:
            cand_mask = cand_mask.contiguous()
        return self.click_scores_indexed(feat, inverse[B * H:], user, B, Cn, cand_mask)

    def backward(self, flat, gflat, dscores, gen=None, table_grad_ready=None):
        """Accumulates every parameter gradient of the saved training forward into gflat (same layout as flat)."""
        sv = self._saved
        if sv is None:
            raise _lib.NrmsError("backward() without a training forward")
        if gen is not None and gen != sv["gen"]:
            raise _lib.NrmsError("backward() of training forward #%d, but the saved activations belong to forward #%d "
                                 "(two training forwards were run before one backward)" % (gen, sv["gen"]))
        d = self.dims
        B, H, Cn, N, F, dt = sv["B"], sv["H"], sv["C"], sv["N"], d.news_feature_size, d.word_embed_size
        p, seed = sv["p"], sv["seed"]
        dfeat = self._buf("d_feat", N * F)[:N * F].view(N, F)
        duser = self._buf("d_user_vec", B * F)[:B * F].view(B, F)
        cand = sv["feat"][B * H:].view(B, Cn, F)
        rc = self.lib.nrms_click_score_bwd(B, Cn, F, _lib.ptr(cand), _lib.ptr(sv["user"]), _lib.ptr(sv["mask"]),
                                           _lib.ptr(dscores.contiguous()), C.c_void_p(dfeat[B * H:].data_ptr()), _lib.ptr(duser),
                                           _stream())
        _lib.check(rc, "nrms_click_score_bwd")
        desc_u = self._desc("user_encoder", B, H, p, seed ^ 0x2545F4914F6CDD1D)
        desc_t = self._desc("news_encoder", N, sv["ids_t"].shape[1], p, seed)
        desc_a = self._desc("news_encoder", N, sv["ids_a"].shape[1], p, seed ^ 0x5DEECE66D1CE4E5B)
        ws = self._bwd_workspace(desc_u, desc_t, desc_a)
        # user encoder -> d(normed history)
        dnormed = self._buf("d_normed", B * H * F)[:B * H * F].view(B * H, F)
        wu, gu = self._ptrs(_lib.EncoderWeights, flat, "user_encoder"), self._ptrs(_lib.EncoderGrads, gflat, "user_encoder")
        rc = self.lib.nrms_encoder_bwd(C.byref(desc_u), C.byref(wu), None, _lib.ptr(sv["normed"]), None,
                                       C.byref(self._acts("user", desc_u, gather=False)), _lib.ptr(duser), C.byref(gu),
                                       _lib.ptr(dnormed), _lib.ptr(ws), C.c_size_t(ws.numel() * 4), _stream())
        _lib.check(rc, "nrms_encoder_bwd(user)")
        # LayerNorm -> d(history feature rows), d(norm.weight), d(norm.bias)
        lnb = int(self.lib.nrms_layernorm_bwd_workspace_bytes(F))
        lws = self._buf("ln_ws", (lnb + 3) // 4)
        rc = self.lib.nrms_layernorm_bwd(C.c_int64(B * H), F, _lib.ptr(sv["feat"]), C.c_void_p(flat.data_ptr() + 4 * self._off("norm.weight")),
                                         _lib.ptr(sv["stats"]), _lib.ptr(dnormed), _lib.ptr(dfeat),
                                         C.c_void_p(gflat.data_ptr() + 4 * self._off("norm.weight")), _lib.ptr(lws),
                                         C.c_size_t(lws.numel() * 4), _stream())
        _lib.check(rc, "nrms_layernorm_bwd")
        # feature rows -> d(title vectors), d(abstract vectors), category tables
        dtv = self._buf("d_title_vec", N * dt)[:N * dt].view(N, dt)
        dav = self._buf("d_abst_vec", N * dt)[:N * dt].view(N, dt)
        f = self._features_desc(flat, N, None, None, sv["categ"], sv["subcateg"], p, seed)
        gb = gflat.data_ptr()
        rc = self.lib.nrms_news_features_bwd(C.byref(f), _lib.ptr(dfeat), _lib.ptr(dtv), _lib.ptr(dav),
                                             C.c_void_p(gb + 4 * self._off("news_encoder.category_embedding.weight")),
                                             C.c_void_p(gb + 4 * self._off("news_encoder.subcategory_embedding.weight")), _stream())
        _lib.check(rc, "nrms_news_features_bwd")
        # the two passes of the word-level encoder accumulate into the same weight and table gradients
        split_t, split_a = sv["split"]
        self.encode_text_backward(flat, gflat, sv["ids_t"], "title", dtv, p, seed, ws, split_t)
        self.encode_text_backward(flat, gflat, sv["ids_a"], "abst", dav, p, seed ^ 0x5DEECE66D1CE4E5B, ws, split_a)
        if table_grad_ready is not None:
            table_grad_ready()
