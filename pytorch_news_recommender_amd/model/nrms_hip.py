"""NRMS on MI355X: drop-in for the reference's ``model.nrms_v0.Model``.

Same constructor (``Model(config)``), same ``forward(batch_dict) -> FloatTensor[B, C]`` of raw
logits with masked candidates at -1e9, same 19 parameter names (so ``state_dict`` files
interchange), same helper API (``get_news_vector`` / ``get_user_vector`` / ``get_prediction``)
as /root/reference/MIND_2020/model/nrms_v0.py:218-312 -- but every op runs in the
hand-written HIP kernels of libnrms_hip.so through the C ABI (include/nrms_hip.h).
There is no PyTorch/CPU fallback: without the library or a GPU, forward raises.

The module tree below only *names and initialises* parameters the way the reference does
(same construction order and initialisers, so ``torch.manual_seed(s)`` gives the same initial
weights); the nn.Linear / nn.Embedding forwards are never called.
"""
from __future__ import annotations

import numpy as np
import os

import torch
import torch.nn as nn

from .. import _lib
from ..engine import FlatLayout, ModelDims, NRMSEngine


class _MultiHeadSelfAttentionParams(nn.Module):
    """Parameter holder mirroring nrms_v0.py:26-44 (W_Q, W_K, W_V Linear(d,d), xavier-uniform weights)."""

    def __init__(self, d_model, num_attention_heads):
        super().__init__()
        assert d_model % num_attention_heads == 0
        self.d_model = d_model
        self.num_attention_heads = num_attention_heads
        self.W_Q = nn.Linear(d_model, d_model)
        self.W_K = nn.Linear(d_model, d_model)
        self.W_V = nn.Linear(d_model, d_model)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=1)


class _AdditiveAttentionParams(nn.Module):
    """Parameter holder mirroring nrms_v0.py:84-93."""

    def __init__(self, query_vector_dim, candidate_vector_dim):
        super().__init__()
        self.linear = nn.Linear(candidate_vector_dim, query_vector_dim)
        self.attention_query_vector = nn.Parameter(torch.empty(query_vector_dim).uniform_(-0.1, 0.1))


class _NewsEncoderParams(nn.Module):
    """nrms_v0.py:130-152: Sequential(Embedding.from_pretrained(freeze=False, padding_idx=0), Dropout)."""

    def __init__(self, config, table):
        super().__init__()
        self.word_embedding = nn.Sequential(
            nn.Embedding.from_pretrained(table, freeze=False, padding_idx=0),
            nn.Dropout(p=config.dropout, inplace=False))
        self.multihead_self_attention = _MultiHeadSelfAttentionParams(config.word_embed_size,
                                                                      config.num_attention_heads)
        self.additive_attention = _AdditiveAttentionParams(config.query_vector_dim, config.word_embed_size)


class _UserEncoderParams(nn.Module):
    """nrms_v0.py:179-186."""

    def __init__(self, config):
        super().__init__()
        self.multihead_self_attention = _MultiHeadSelfAttentionParams(config.word_embed_size,
                                                                      config.num_attention_heads)
        self.additive_attention = _AdditiveAttentionParams(config.query_vector_dim, config.word_embed_size)


def _load_table(config, pretrained_word_embedding):
    if pretrained_word_embedding is None:
        path = config.data_path + config.word_embedding_pretrained        # nrms_v0.py:134-135
        arr = np.load(path)["embeddings"].astype("float32")
        return torch.tensor(arr)
    return torch.as_tensor(np.asarray(pretrained_word_embedding, dtype=np.float32)).clone()


def _ids_on(dev, x):
    """Word ids on the device: int32 feeds stay int32 over PCIe (the library validates either width into its own
    int64 copy), anything else becomes the reference's int64."""
    t = torch.as_tensor(x)
    return t.to(dev, dtype=torch.int32 if t.dtype == torch.int32 else torch.int64, non_blocking=True)


class _NRMSFunction(torch.autograd.Function):
    """scores = NRMS(batch; params) with the backward in HIP (autograd sees one node)."""

    @staticmethod
    def forward(ctx, model, bt, ct, mask, p_drop, seed, *params):
        ctx.model = model
        scores = model._engine.forward(model._flat, bt, ct, mask, training=True, p_drop=p_drop, seed=seed)
        ctx.gen = model._engine._saved["gen"]
        return scores

    @staticmethod
    def backward(ctx, dscores):
        model = ctx.model
        # Default: a fresh flat gradient buffer per backward -- autograd adopts the returned views as .grad, and a tensor
        # the caller kept from an earlier step (a saved p.grad, a hook's argument) must keep its values, as torch guarantees.
        # model.reuse_grad_buffer = True (set by train_eval.train for the reference's loop, which zeroes the gradients
        # before every backward, train_eval.py:115, and keeps nothing) opts into ONE persistent 57.6 MB buffer instead;
        # even then a fresh buffer is used while gradients are being accumulated (.grad still set).
        reuse = bool(getattr(model, "reuse_grad_buffer", False)) and not any(p.grad is not None for p in model.parameters())
        gflat = model._autograd_grad if reuse else None
        if gflat is None or gflat.shape != model._flat.shape or gflat.device != model._flat.device:
            gflat = torch.empty_like(model._flat)
            if reuse:
                model._autograd_grad = gflat
        gflat.zero_()
        model._engine.backward(model._flat, gflat, dscores, gen=ctx.gen)
        if model._engine.precision == "fp16":
            # the caller's own optimizer follows (torch.optim.Adam, train_eval.py:127): inf / nan elements of an overflowing
            # fp16 backward become 0 and are counted; the engine lowers its loss scale when the count reaches the host
            model._engine.grad_guard(gflat)
            model._engine.note_grad_check()
        grads = tuple(model._layout.view(gflat, n) for n in model._names)
        return (None, None, None, None, None, None) + grads


class Model(nn.Module):
    """NRMS network: 1+K candidate titles and the clicked-title history -> click logits."""

    def __init__(self, config, pretrained_word_embedding=None):
        super().__init__()
        self.config = config
        table = _load_table(config, pretrained_word_embedding)
        V, d = table.shape
        if d != config.word_embed_size:
            raise ValueError("embedding width %d != config.word_embed_size %d" % (d, config.word_embed_size))
        self._build_modules(config, table)
        self._dims = self._make_dims(config, int(V), int(d))
        self._layout = self._make_layout(self._dims)
        self._names = self._layout.names
        named = dict(self.named_parameters())
        missing = [n for n in self._names if n not in named]
        assert not missing and len(named) == len(self._names), (missing, sorted(named))
        self._flat = None
        self._pad_zero = None
        self._engine = None
        self._opt = None
        self._calls = 0
        self._autograd_grad = None
        self.reuse_grad_buffer = False      # autograd path: see _NRMSFunction.backward
        self._flatten(table.device)
        # a load through ANY parent (the dispatch wrapper of model/__init__.py included) may change the embedding
        # table: nn.Module.load_state_dict recurses via _load_from_state_dict and never calls a child's
        # load_state_dict, so the reset lives in a post-hook, which fires on recursive loads too
        self.register_load_state_dict_post_hook(Model._reset_pad_flag)

    # ---- topology hooks (overridden by model/nrms_v1_hip.py) ---------------------------------------
    def _build_modules(self, config, table):
        self.news_encoder = _NewsEncoderParams(config, table)
        self.user_encoder = _UserEncoderParams(config)

    def _make_dims(self, config, V, d):
        return ModelDims(n_words=V, word_embed_size=d, num_attention_heads=int(config.num_attention_heads),
                         query_vector_dim=int(config.query_vector_dim))

    def _make_layout(self, dims):
        return FlatLayout(dims)

    def _make_engine(self, device, precision):
        return NRMSEngine(self._dims, device, precision=precision)

    # ---- flat parameter storage ----------------------------------------------------------
    def _flatten(self, device):
        """(Re)build the flat parameter buffer on `device` and point every Parameter at its slice."""
        named = dict(self.named_parameters())
        flat = torch.empty(self._layout.total, dtype=torch.float32, device=device)
        for n in self._names:
            v = self._layout.view(flat, n)
            v.copy_(named[n].data)
            named[n].data = v
        self._flat = flat
        self._opt = None
        self._pad_zero = None

    @staticmethod
    def _reset_pad_flag(module, incompatible_keys):
        module._pad_zero = None                    # the embedding table may have changed

    def refresh_pad_row_flag(self):
        """Re-evaluate NRMS_FLAG_PAD_ROW_ZERO (include/nrms_hip.h) after writing into the embedding table by
        hand.  Training never needs it: row 0 has an identically zero gradient (padding_idx), so Adam leaves
        it where it was when the weights were loaded."""
        self._pad_zero = None

    def _views_intact(self):
        base = self._flat.data_ptr()
        named = dict(self.named_parameters())
        for n in self._names:
            off = self._layout.entries[n][0]
            p = named[n]
            if p.data_ptr() != base + 4 * off or p.device != self._flat.device:
                return False
        return True

    def _prepare(self):
        """Make sure parameters live in one flat GPU buffer (``.to(device)`` replaces tensors)."""
        dev = next(self.parameters()).device
        if not self._views_intact():
            self._flatten(dev)
        if self._flat.device.type != "cuda":
            raise _lib.NrmsError("NRMS HIP model parameters are on %s: move the model to a GPU "
                                 "(there is no CPU fallback)" % self._flat.device)
        prec = getattr(self.config, "precision", "fp32")
        if self._engine is None or self._engine.device != self._flat.device:
            self._engine = self._make_engine(self._flat.device, prec)
        elif self._engine.precision != prec:
            self._engine.set_precision(prec)
        self._prepare_calls = getattr(self, "_prepare_calls", 0) + 1
        if getattr(self, "_pad_zero", None) is None or self._prepare_calls % 256 == 0:
            # one host sync per weight load (and a cheap re-validation every 256 calls, should somebody write
            # into the table by hand without refresh_pad_row_flag()): is the padding row all zeros?
            tname = [n for n in self._names if n.endswith("word_embedding.0.weight") or n.endswith("word_embedding.weight")][0]
            self._pad_zero = bool((self._layout.view(self._flat, tname)[0] == 0).all().item())
        self._engine.fp16_user_encoder = bool(getattr(self.config, "fp16_user_encoder", False))
        self._engine.fp16_inference = bool(getattr(self.config, "fp16_inference", False))
        self._engine.fp16_wide_heads = bool(getattr(self.config, "fp16_v1_news_encoder", False))
        self._engine.pad_row_zero = self._pad_zero and bool(getattr(self.config, "skip_padding_tokens", True))
        return self._flat.device

    def _next_seed(self):
        self._calls += 1
        return (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + self._calls * 0xD1B54A32D192ED03
                + getattr(self, "_rank_salt", 0)) & 0xFFFFFFFFFFFFFFFF

    # ---- reference API ----------------------------------------------------------------------
    def forward(self, batch):
        """batch: the collated dict of data_handler.MyDataset (CPU or GPU tensors); only
        'browsed_titles' [B,H,L], 'candidate_titles' [B,C,L] and 'candidate_mask' [B,C] are read
        (nrms_v0.py:248,250,272).  Returns click logits [B,C] on the GPU."""
        dev = self._prepare()
        bt, ct = _ids_on(dev, batch["browsed_titles"]), _ids_on(dev, batch["candidate_titles"])
        mask = batch.get("candidate_mask") if hasattr(batch, "get") else batch["candidate_mask"]
        if mask is not None:
            mask = torch.as_tensor(mask).to(dev, dtype=torch.uint8, non_blocking=True)
        p_drop = float(self.config.dropout) if self.training else 0.0
        seed = self._next_seed() if p_drop > 0 else 0
        params = [p for _, p in self._ordered_params()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _NRMSFunction.apply(self, bt, ct, mask, p_drop, seed, *params)
        if p_drop == 0.0 and getattr(self, "dedup_inference", True):
            get = batch.get if hasattr(batch, "get") else (lambda k: None)
            bn, cn = get("browsed_ids"), get("candidate_ids")
            if self._engine._news_cache is not None and bn is not None and cn is not None:
                # inside train_eval.evaluate / test: news vectors are cached across batches by news id
                return self._engine.forward_cached(self._flat, bt, ct, torch.as_tensor(bn).to(dev), torch.as_tensor(cn).to(dev), mask)
            scores, self.last_unique_titles = self._engine.forward_dedup(self._flat, bt, ct, mask)
            return scores
        return self._engine.forward(self._flat, bt, ct, mask, training=False, p_drop=p_drop, seed=seed)

    def _ordered_params(self):
        named = dict(self.named_parameters())
        return [(n, named[n]) for n in self._names]

    def get_news_vector(self, news):
        """news [N, L] title ids -> [N, d] (nrms_v0.py:278-289); inference only."""
        dev = self._prepare()
        ids = _ids_on(dev, news)
        p_drop = float(self.config.dropout) if self.training else 0.0
        p_embed = 0.0 if self._dims.style == "v1" else p_drop
        return self._engine.encode_titles(self._flat, ids, p_embed=p_embed, p_ctx=p_drop,
                                          seed=self._next_seed() if p_drop else 0)

    def get_user_vector(self, clicked_news_vector):
        """[B, H, d] -> [B, d] (nrms_v0.py:291-299); inference only."""
        dev = self._prepare()
        x = torch.as_tensor(clicked_news_vector).to(dev, dtype=torch.float32).contiguous()
        return self._engine.encode_users(self._flat, x)

    def get_prediction(self, news_vector, user_vector):
        """news_vector [C, d], user_vector [d] -> [C] (nrms_v0.py:301-312)."""
        dev = self._prepare()
        nv = torch.as_tensor(news_vector).to(dev, dtype=torch.float32).contiguous().unsqueeze(0)
        uv = torch.as_tensor(user_vector).to(dev, dtype=torch.float32).contiguous().unsqueeze(0)
        return self._engine.click_scores(nv, uv).squeeze(0)

    # ---- fused training step (the build's own loop; same math as train_eval.py:111-127) -----
    def train_step(self, batch, lr=None, betas=(0.9, 0.999), eps=1e-8, world_size=1, all_reduce=None,
                   global_batch=None):
        """forward + CE(label 0) + backward + [gradient all-reduce] + Adam, all in HIP on flat
        buffers, no host sync.  Returns the local loss SUM over the batch as a device scalar
        (divide by the batch size for the reference's mean loss).

        all_reduce: callable(flat_grad_tensor) that sums gradients over data-parallel ranks
        (RCCL); gradients are scaled by 1/global_batch so the summed result is the gradient of
        the mean loss over the global batch."""
        dev = self._prepare()
        eng = self._engine
        bt, ct = _ids_on(dev, batch["browsed_titles"]), _ids_on(dev, batch["candidate_titles"])
        mask = batch.get("candidate_mask")
        if mask is not None:
            mask = torch.as_tensor(mask).to(dev, dtype=torch.uint8, non_blocking=True)
        if self._opt is None:
            self._opt = dict(step=0, g=torch.zeros_like(self._flat), m=torch.zeros_like(self._flat),
                             v=torch.zeros_like(self._flat))
        st = self._opt
        p_drop = float(self.config.dropout) if self.training else 0.0
        seed = self._next_seed() if p_drop > 0 else 0
        B = bt.shape[0]
        gb = B * world_size if global_batch is None else global_batch
        scores = eng.forward(self._flat, bt, ct, mask, training=True, p_drop=p_drop, seed=seed)
        loss_sum, dscores = eng.ce_loss(scores, grad_scale=1.0 / gb)
        st["g"].zero_()
        lr_ = float(self.config.learning_rate if lr is None else lr)
        overlap = not os.environ.get("NRMS_NO_OVERLAP")
        if all_reduce is not None and hasattr(all_reduce, "owned"):
            # parallel.ShardedGradSync: reduce-scatter (the table region underneath the deferred weight-gradient GEMMs),
            # Adam on the 1/world of the parameters this rank owns, all-gather of the updated parameters
            pending = []
            if overlap:
                eng.backward(self._flat, st["g"], dscores, table_grad_ready=lambda: pending.append(all_reduce.start(st["g"], 0)))
            else:
                eng.backward(self._flat, st["g"], dscores)
                pending.append(all_reduce.start(st["g"], 0))
            pending.append(all_reduce.start(st["g"], 1))
            for h in pending:
                h.wait()
            st["step"] += 1
            for lo, hi, gshard in all_reduce.owned():
                eng.adam_step(self._flat[lo:hi], gshard, st["m"][lo:hi], st["v"][lo:hi], st["step"], lr=lr_, betas=betas, eps=eps)
            if eng.precision == "fp16":
                eng.note_grad_check()
            all_reduce.gather(self._flat)
            self._last_scores = scores
            return loss_sum
        if all_reduce is not None and hasattr(all_reduce, "start") and overlap:
            # the table gradient (first V*d floats of the flat buffer, 95 % of the bytes) is reduced underneath
            # the deferred d(W_qkv) GEMM; the remaining 2.6 MB follow when the backward has been enqueued
            n_table = self._dims.n_words * self._dims.word_embed_size
            pending = []
            eng.backward(self._flat, st["g"], dscores,
                         table_grad_ready=lambda: pending.append(all_reduce.start(st["g"][:n_table])))
            all_reduce(st["g"][n_table:])
            for h in pending:
                h.wait()
        else:
            eng.backward(self._flat, st["g"], dscores)
            if all_reduce is not None:
                all_reduce(st["g"])
        st["step"] += 1
        eng.adam_step(self._flat, st["g"], st["m"], st["v"], st["step"], lr=lr_, betas=betas, eps=eps)
        if eng.precision == "fp16":
            eng.note_grad_check()
        self._last_scores = scores
        return loss_sum

    @property
    def engine(self):
        self._prepare()
        return self._engine
