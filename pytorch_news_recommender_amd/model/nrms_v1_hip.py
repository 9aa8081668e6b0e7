"""NRMS, ``nrms_v1`` topology, on MI355X: multi-head self-attention WITH the output projection
``output_linear`` (W_O), ``title_heads_num`` heads in the news encoder and ``num_attention_heads`` in
the user encoder, dropout only after the attention block -- /root/reference/MIND_2020/model/nrms_v1.py:
41-80 (MHSA), 82-105 (additive attention), 109-162 (NewsEncoder), 199-211 (UserEncoder), 242-294 (Model.forward).

Parameter names are those of the reference's classes (``linear_layers.{0,1,2}``, ``output_linear``,
``query_vector``).  The reference's own ``nrms_v1.Model(config)`` cannot be constructed as committed
(``UserEncoder`` passes 2 arguments to a 3-argument constructor, nrms_v1.py:203-204 vs :46), so there is no
checkpoint to interchange with; the forward semantics are pinned through its working classes
(fixture g3) and the oracle.  Like the reference's ``forward`` (nrms_v1.py:286), no mask is applied in the
model; the masked primitives (pairwise attention mask, masked additive attention) are available through
``engine.encode_titles(..., mask=, mask_mode=)`` / ``encode_users``.
"""
import torch
import torch.nn as nn

from ..engine import ModelDims
from . import nrms_hip


class _MultiHeadSelfAttentionParams(nn.Module):
    """nrms_v1.py:46-64: three Linear(d,d) in a ModuleList + output_linear, xavier-uniform weights."""

    def __init__(self, h, d_model):
        super().__init__()
        assert d_model % h == 0
        self.h = h
        self.linear_layers = nn.ModuleList([nn.Linear(d_model, d_model) for _ in range(3)])
        self.output_linear = nn.Linear(d_model, d_model)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=1)


class _AdditiveAttentionParams(nn.Module):
    """nrms_v1.py:83-87."""

    def __init__(self, query_vector_dim, input_vector_dim):
        super().__init__()
        self.linear = nn.Linear(input_vector_dim, query_vector_dim)
        self.query_vector = nn.Parameter(torch.empty(query_vector_dim).uniform_(-0.1, 0.1))


class _NewsEncoderParams(nn.Module):
    """nrms_v1.py:110-125."""

    def __init__(self, config, table):
        super().__init__()
        self.word_embedding = nn.Embedding.from_pretrained(table, freeze=False, padding_idx=0)
        self.multi_head_self_attention = _MultiHeadSelfAttentionParams(config.title_heads_num, config.word_embed_size)
        self.additive_attention = _AdditiveAttentionParams(config.query_vector_dim, config.word_embed_size)
        self.dropout = nn.Dropout(config.dropout)


class _UserEncoderParams(nn.Module):
    """nrms_v1.py:201-206 (with the constructor call the reference meant)."""

    def __init__(self, config):
        super().__init__()
        self.multi_head_self_attention = _MultiHeadSelfAttentionParams(config.num_attention_heads,
                                                                       config.word_embed_size)
        self.additive_attention = _AdditiveAttentionParams(config.query_vector_dim, config.word_embed_size)


class Model(nrms_hip.Model):
    def _build_modules(self, config, table):
        self.news_encoder = _NewsEncoderParams(config, table)
        self.user_encoder = _UserEncoderParams(config)

    def _make_dims(self, config, V, d):
        return ModelDims(n_words=V, word_embed_size=d, num_attention_heads=int(config.num_attention_heads),
                         query_vector_dim=int(config.query_vector_dim), news_heads=int(config.title_heads_num),
                         output_proj=True, style="v1")
