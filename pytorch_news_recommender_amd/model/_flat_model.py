"""Shared host plumbing of the models that have no counterpart in the reference (model/hierec_hip.py,
model/graph_hip.py): parameters as views of one flat fp32 buffer (one Adam launch, one gradient all-reduce), the NRMS news
encoder's parameter names (model/nrms_v0.py:130-152), and the reference's plugin contract -- ``Model(config)``,
``forward(batch_dict) -> FloatTensor[B, C]`` (model/__init__.py:22-23,38) -- around an engine with
``forward(flat, batch, training, p_drop, seed)`` / ``backward(flat, gflat, dscores, gen)``.  No CPU fallback."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..engine import ModelDims


class FlatLayout2:
    """Flat parameter buffer: the NRMS news encoder's ten tensors (table first, W_Q | W_K | W_V adjacent as the kernels need),
    then the model's own tensors in the order given: extra = [(name, shape), ...]."""

    def __init__(self, dims: ModelDims, extra):
        V, d, q = dims.n_words, dims.word_embed_size, dims.query_vector_dim
        if d % 4 or q % 4:
            raise ValueError("word_embed_size and query_vector_dim must be multiples of 4")
        self.dims = dims
        self.entries, self.blocks = {}, {"news_encoder": {}}
        off = 0

        def add(name, shape, role=None):
            nonlocal off
            n = int(np.prod(shape))
            self.entries[name] = (off, tuple(shape), n)
            if role is not None:
                self.blocks["news_encoder"][role] = off
            off += n

        a = "news_encoder.multihead_self_attention."
        add("news_encoder.word_embedding.0.weight", (V, d), "table")
        for nm, r in zip(("W_Q", "W_K", "W_V"), ("wq", "wk", "wv")):
            add(a + nm + ".weight", (d, d), r)
        for nm, r in zip(("W_Q", "W_K", "W_V"), ("bq", "bk", "bv")):
            add(a + nm + ".bias", (d,), r)
        add("news_encoder.additive_attention.linear.weight", (q, d), "wa")
        add("news_encoder.additive_attention.linear.bias", (q,), "ba")
        add("news_encoder.additive_attention.attention_query_vector", (q,), "qv")
        for name, shape in extra:
            add(name, shape)
        self.total = off
        self.names = list(self.entries)
        self.table = 0

    def view(self, flat, name):
        off, shp, n = self.entries[name]
        return flat[off:off + n].view(shp)


class AdditiveParams(nn.Module):
    """The reference's AdditiveAttention parameters (model/nrms_v0.py:84-93): Linear(d, q) + a query vector U(-0.1, 0.1)."""

    def __init__(self, q, d):
        super().__init__()
        self.linear = nn.Linear(d, q)
        self.attention_query_vector = nn.Parameter(torch.empty(q).uniform_(-0.1, 0.1))


def additive_entries(prefix, q, d):
    return [(prefix + ".linear.weight", (q, d)), (prefix + ".linear.bias", (q,)), (prefix + ".attention_query_vector", (q,))]


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, p_drop, seed, *params):
        ctx.model = model
        scores = model._engine.forward(model._flat, batch, training=True, p_drop=p_drop, seed=seed)
        ctx.gen = model._engine._saved["gen"]
        return scores

    @staticmethod
    def backward(ctx, dscores):
        model = ctx.model
        gflat = torch.zeros_like(model._flat)
        model._engine.backward(model._flat, gflat, dscores, gen=ctx.gen)
        model._zero_frozen_rows(gflat)
        if model._engine.precision == "fp16":
            model._engine.grad_guard(gflat)
            model._engine.note_grad_check()
        return (None, None, None, None) + tuple(model._layout.view(gflat, n) for n in model._names)


class FlatHipModel(nn.Module):
    """Subclasses create their nn.Parameters, then call ``_finish(layout)``; they provide ``KEYS`` (batch-dict keys, optional ones
    in ``OPTIONAL``), ``_make_engine(device, precision)`` and optionally ``_zero_frozen_rows(gflat)`` (padding_idx rows)."""
    KEYS = ()
    OPTIONAL = ("candidate_mask",)

    def _finish(self, layout, device):
        self._layout = layout
        self._names = layout.names
        named = dict(self.named_parameters())
        assert sorted(named) == sorted(self._names), sorted(set(named) ^ set(self._names))
        self._flat = self._engine = self._opt = None
        self._pad_zero = None
        self._calls = 0
        self._flatten(device)

    def _flatten(self, device):
        named = dict(self.named_parameters())
        flat = torch.empty(self._layout.total, dtype=torch.float32, device=device)
        for n in self._names:
            v = self._layout.view(flat, n)
            v.copy_(named[n].data)
            named[n].data = v
        self._flat, self._opt, self._pad_zero = flat, None, None

    def _zero_frozen_rows(self, gflat):
        pass

    def _prepare(self):
        dev = next(self.parameters()).device
        named = dict(self.named_parameters())
        base = self._flat.data_ptr()
        if any(named[n].data_ptr() != base + 4 * self._layout.entries[n][0] or named[n].device != self._flat.device for n in self._names):
            self._flatten(dev)
        if self._flat.device.type != "cuda":
            raise _lib.NrmsError("%s: parameters are on %s; move the model to a GPU (there is no CPU fallback)" % (type(self).__module__, self._flat.device))
        prec = getattr(self.config, "precision", "fp32")
        if self._engine is None or self._engine.device != self._flat.device:
            self._engine = self._make_engine(self._flat.device, prec)
        elif self._engine.precision != prec:
            self._engine.set_precision(prec)
        if self._pad_zero is None:
            self._pad_zero = bool((self._layout.view(self._flat, "news_encoder.word_embedding.0.weight")[0] == 0).all().item())
        self._engine.fp16_inference = bool(getattr(self.config, "fp16_inference", False))
        self._engine.pad_row_zero = self._pad_zero and bool(getattr(self.config, "skip_padding_tokens", True))
        return self._flat.device

    def _next_seed(self):
        self._calls += 1
        # (_rank_salt: set by run_v0 per data-parallel rank, so that the ranks draw different dropout masks)
        return (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + self._calls * 0xD1B54A32D192ED03 + getattr(self, "_rank_salt", 0)) & 0xFFFFFFFFFFFFFFFF

    def _device_batch(self, batch, dev):
        out = {}
        for k in self.KEYS:
            v = batch.get(k) if hasattr(batch, "get") else batch[k]
            if v is None:
                if k in self.OPTIONAL:
                    continue
                raise KeyError("%s: the batch dict lacks %r" % (type(self).__module__, k))
            out[k] = torch.as_tensor(v).to(dev, non_blocking=True)
        return out

    def forward(self, batch):
        dev = self._prepare()
        b = self._device_batch(batch, dev)
        p_drop = float(self.config.dropout) if self.training else 0.0
        seed = self._next_seed() if p_drop > 0 else 0
        named = dict(self.named_parameters())
        params = [named[n] for n in self._names]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _Fn.apply(self, b, p_drop, seed, *params)
        return self._engine.forward(self._flat, b, training=False, p_drop=p_drop, seed=seed)

    def train_step(self, batch, lr=None, betas=(0.9, 0.999), eps=1e-8, world_size=1, all_reduce=None, global_batch=None):
        """forward + CE(label 0) + backward + [gradient all-reduce over the data-parallel ranks] + Adam on the flat buffers (same
        math as train_eval.py:111-127).  Returns the local loss SUM over the batch as a device scalar."""
        dev = self._prepare()
        eng = self._engine
        b = self._device_batch(batch, dev)
        if self._opt is None:
            self._opt = dict(step=0, g=torch.zeros_like(self._flat), m=torch.zeros_like(self._flat), v=torch.zeros_like(self._flat))
        st = self._opt
        p_drop = float(self.config.dropout) if self.training else 0.0
        seed = self._next_seed() if p_drop > 0 else 0
        scores = eng.forward(self._flat, b, training=True, p_drop=p_drop, seed=seed)
        gb = scores.shape[0] * world_size if global_batch is None else global_batch
        loss_sum, dscores = eng.ce_loss(scores, grad_scale=1.0 / gb)
        st["g"].zero_()
        eng.backward(self._flat, st["g"], dscores)
        self._zero_frozen_rows(st["g"])
        if all_reduce is not None:
            all_reduce(st["g"])
        st["step"] += 1
        eng.adam_step(self._flat, st["g"], st["m"], st["v"], st["step"], lr=float(self.config.learning_rate if lr is None else lr), betas=betas, eps=eps)
        if eng.precision == "fp16":
            eng.note_grad_check()
        return loss_sum

    @property
    def engine(self):
        self._prepare()
        return self._engine
