"""``nrms_naml`` on MI355X: NRMS over title + abstract + category / sub-category embeddings with a
LayerNorm on the history and an 800-wide user encoder -- /root/reference/MIND_2020/model/nrms_naml.py:
14-41 (attention with dropout on the probabilities), 42-75 (MHSA with output_linear), 77-100 (additive attention),
103-177 (NewsEncoder), 179-191 (UserEncoder), 196-257 (Model).

Same plugin contract as the other models (``model.nrms_naml_hip.Model(config)``, model/__init__.py:22-23); parameter names
and ``state_dict()`` order are the reference's, so checkpoints interchange with ``model.nrms_naml``.  The batch keys read
are the reference's (nrms_naml.py:217-228,245): ``browsed_titles / _absts / _categ_ids / _subcateg_ids``, the four
``candidate_*`` counterparts and ``candidate_mask`` -- all emitted by data_handler.MyDataset.
"""
import torch
import torch.nn as nn

from .. import _lib
from ..naml_engine import NamlDims, NamlEngine, NamlLayout
from . import nrms_hip

ID_KEYS = ("browsed_titles", "browsed_absts", "browsed_categ_ids", "browsed_subcateg_ids",
           "candidate_titles", "candidate_absts", "candidate_categ_ids", "candidate_subcateg_ids")


class _MultiHeadSelfAttentionParams(nn.Module):
    """nrms_naml.py:47-59: three Linear(d, d) in a ModuleList + output_linear, torch's default initialisation."""

    def __init__(self, h, d_model):
        super().__init__()
        assert d_model % h == 0
        self.h = h
        self.linear_layers = nn.ModuleList([nn.Linear(d_model, d_model) for _ in range(3)])
        self.output_linear = nn.Linear(d_model, d_model)


class _AdditiveAttentionParams(nn.Module):
    """nrms_naml.py:78-82."""

    def __init__(self, query_vector_dim, input_vector_dim):
        super().__init__()
        self.linear = nn.Linear(input_vector_dim, query_vector_dim)
        self.query_vector = nn.Parameter(torch.empty(query_vector_dim).uniform_(-0.1, 0.1))


class _NewsEncoderParams(nn.Module):
    """nrms_naml.py:104-119."""

    def __init__(self, config, table):
        super().__init__()
        self.category_embedding = nn.Embedding(config.category_nums, config.cate_embed_size, padding_idx=0)
        self.subcategory_embedding = nn.Embedding(config.subcategory_nums, config.cate_embed_size, padding_idx=0)
        self.word_embedding = nn.Embedding.from_pretrained(table, freeze=False, padding_idx=0)
        self.multi_head_self_attention = _MultiHeadSelfAttentionParams(config.title_heads_num, config.word_embed_size)
        self.additive_attention = _AdditiveAttentionParams(config.query_vector_dim, config.word_embed_size)


class _UserEncoderParams(nn.Module):
    """nrms_naml.py:181-186."""

    def __init__(self, config):
        super().__init__()
        self.multi_head_self_attention = _MultiHeadSelfAttentionParams(config.user_heads_num, config.news_feature_size)
        self.additive_attention = _AdditiveAttentionParams(config.query_vector_dim_large, config.news_feature_size)


class _NamlFunction(torch.autograd.Function):
    """scores = nrms_naml(batch; params) with the backward in HIP (autograd sees one node)."""

    @staticmethod
    def forward(ctx, model, ids, mask, p_drop, seed, *params):
        ctx.model = model
        scores = model._engine.forward(model._flat, ids, mask, training=True, p_drop=p_drop, seed=seed)
        ctx.gen = model._engine._saved["gen"]
        return scores

    @staticmethod
    def backward(ctx, dscores):
        model = ctx.model
        # fresh buffer per backward unless the caller opted into reuse (model.reuse_grad_buffer: nrms_hip._NRMSFunction)
        reuse = bool(getattr(model, "reuse_grad_buffer", False)) and not any(p.grad is not None for p in model.parameters())
        gflat = model._autograd_grad if reuse else None
        if gflat is None or gflat.shape != model._flat.shape or gflat.device != model._flat.device:
            gflat = torch.empty_like(model._flat)
            if reuse:
                model._autograd_grad = gflat
        gflat.zero_()
        model._engine.backward(model._flat, gflat, dscores, gen=ctx.gen)
        grads = tuple(model._layout.view(gflat, n) for n in model._names)
        return (None, None, None, None, None) + grads


class Model(nrms_hip.Model):
    def _build_modules(self, config, table):
        if int(config.news_feature_size) != 2 * int(config.word_embed_size) + 2 * int(config.cate_embed_size):
            raise ValueError("news_feature_size %d != 2 * word_embed_size + 2 * cate_embed_size (nrms_naml.py:174 concatenates "
                             "[title | abstract | category | sub-category])" % config.news_feature_size)
        self.news_encoder = _NewsEncoderParams(config, table)
        self.user_encoder = _UserEncoderParams(config)
        self.norm = nn.LayerNorm(config.news_feature_size)

    def _make_dims(self, config, V, d):
        return NamlDims(n_words=V, word_embed_size=d, title_heads_num=int(config.title_heads_num),
                        query_vector_dim=int(config.query_vector_dim), category_nums=int(config.category_nums),
                        subcategory_nums=int(config.subcategory_nums), cate_embed_size=int(config.cate_embed_size),
                        user_heads_num=int(config.user_heads_num), query_vector_dim_large=int(config.query_vector_dim_large))

    def _make_layout(self, dims):
        return NamlLayout(dims)

    def _make_engine(self, device, precision):
        return NamlEngine(self._dims, device, precision=precision)

    def _inputs(self, batch, dev):
        ids = {k: torch.as_tensor(batch[k]).to(dev, dtype=torch.int64, non_blocking=True) for k in ID_KEYS}
        mask = batch.get("candidate_mask") if hasattr(batch, "get") else batch["candidate_mask"]
        if mask is not None:
            mask = torch.as_tensor(mask).to(dev, dtype=torch.uint8, non_blocking=True)
        return ids, mask

    def forward(self, batch):
        """batch: the collated dict of data_handler.MyDataset.  Returns click logits [B, C] on the GPU."""
        dev = self._prepare()
        ids, mask = self._inputs(batch, dev)
        p_drop = float(self.config.dropout) if self.training else 0.0
        seed = self._next_seed() if p_drop > 0 else 0
        params = [p for _, p in self._ordered_params()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _NamlFunction.apply(self, ids, mask, p_drop, seed, *params)
        # evaluation: every distinct news item of the batch is encoded once (model.dedup_inference = False: every slot)
        self._engine.dedup_inference = bool(getattr(self, "dedup_inference", True))
        return self._engine.forward(self._flat, ids, mask, training=False, p_drop=p_drop, seed=seed)

    def get_news_vector(self, *a, **k):
        raise _lib.NrmsError("nrms_naml has no get_news_vector / get_user_vector / get_prediction (nrms_naml.py:196-257)")

    get_user_vector = get_prediction = get_news_vector

    def train_step(self, batch, lr=None, betas=(0.9, 0.999), eps=1e-8, world_size=1, all_reduce=None, global_batch=None):
        """forward + CE(label 0) + backward + [gradient all-reduce] + Adam on flat buffers (train_eval.py:111-127)."""
        dev = self._prepare()
        eng = self._engine
        ids, mask = self._inputs(batch, dev)
        if self._opt is None:
            self._opt = dict(step=0, g=torch.zeros_like(self._flat), m=torch.zeros_like(self._flat),
                             v=torch.zeros_like(self._flat))
        st = self._opt
        p_drop = float(self.config.dropout) if self.training else 0.0
        seed = self._next_seed() if p_drop > 0 else 0
        B = ids["browsed_titles"].shape[0]
        gb = B * world_size if global_batch is None else global_batch
        scores = eng.forward(self._flat, ids, mask, training=True, p_drop=p_drop, seed=seed)
        loss_sum, dscores = eng.ce_loss(scores, grad_scale=1.0 / gb)
        st["g"].zero_()
        eng.backward(self._flat, st["g"], dscores)
        if all_reduce is not None:
            all_reduce(st["g"])
        st["step"] += 1
        eng.adam_step(self._flat, st["g"], st["m"], st["v"], st["step"],
                      lr=float(self.config.learning_rate if lr is None else lr), betas=betas, eps=eps)
        self._last_scores = scores
        return loss_sum
