"""CPU restatement of nrms_naml -- TEST INFRASTRUCTURE, never imported by the product path.

Plain torch-CPU functional code over a dict of tensors with the reference's own parameter names
(``Model.state_dict()`` of /root/reference/MIND_2020/model/nrms_naml.py).  Pinned by ``tests/golden/g7_naml.npz``
(outputs of the imported reference, ``tests/golden/gen_golden.py:gen_g7``) in ``tests/test_oracle_golden.py``.

What the reference computes, by line:
  Attention.forward            nrms_naml.py:20-41   softmax(Q K^T / sqrt(d_k)) -> DROPOUT ON THE PROBABILITIES -> . V
                                                   (no mask is ever passed on this path: :156,165,189,241)
  MultiHeadSelfAttention       nrms_naml.py:42-75   three Linear(d,d), heads, attention, concat, output_linear
  AdditiveAttention            nrms_naml.py:77-100  tanh(Linear) . query_vector, softmax over the sequence, weighted sum
  NewsEncoder.forward          nrms_naml.py:121-177 title and abstract through the SAME embedding / MHSA / additive
                                                   weights (no embedding dropout), category and sub-category embeddings
                                                   (padding_idx 0), concat [title|abstract|category|subcategory], dropout
  UserEncoder.forward          nrms_naml.py:188-191 MHSA + additive attention over the history, no mask
  Model.forward                nrms_naml.py:216-257 candidates and history through the news encoder, LayerNorm on the
                                                   history only, user vector, dot-product scores, masked_fill(-1e9)
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .nrms_oracle import to_torch, loss_fn  # noqa: F401  (same CE with label 0, train_eval.py:113-118)


def _keep(x, keep, p_drop):
    if keep is None or p_drop == 0.0:
        return x
    return x * keep.to(x.dtype) / (1.0 - p_drop)


def mhsa(p, prefix, X, n_heads, p_drop=0.0, keep_attn=None):
    """nrms_naml.py:42-75 with Attention.forward :20-41.  keep_attn [N, h, S, S] (or None)."""
    N, S, d = X.shape
    dk = d // n_heads

    def proj(i):
        y = F.linear(X, p[prefix + "linear_layers.%d.weight" % i], p[prefix + "linear_layers.%d.bias" % i])
        return y.view(N, S, n_heads, dk).transpose(1, 2)

    q, k, v = proj(0), proj(1), proj(2)
    attn = F.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk), dim=-1)
    attn = _keep(attn, keep_attn, p_drop)
    x = torch.matmul(attn, v).transpose(1, 2).contiguous().view(N, S, d)
    return F.linear(x, p[prefix + "output_linear.weight"], p[prefix + "output_linear.bias"])


def additive(p, prefix, X):
    """nrms_naml.py:77-100 (mask is None on this path)."""
    t = torch.tanh(F.linear(X, p[prefix + "linear.weight"], p[prefix + "linear.bias"]))
    w = F.softmax(torch.matmul(t, p[prefix + "query_vector"]), dim=1)
    return torch.bmm(w.unsqueeze(1), X).squeeze(1)


def text_vector(p, ids, n_heads, p_drop=0.0, keep_attn=None):
    """One slot loop body of NewsEncoder.forward (:154-158 / :163-167), for all slots at once: ids [N, L] -> [N, d]."""
    x = F.embedding(ids, p["news_encoder.word_embedding.weight"], padding_idx=0)        # :111-113
    y = mhsa(p, "news_encoder.multi_head_self_attention.", x, n_heads, p_drop, keep_attn)
    return additive(p, "news_encoder.additive_attention.", y)


def news_features(p, titles, absts, categ, subcateg, n_heads, p_drop=0.0, keep=None):
    """NewsEncoder.forward: [N, Lt], [N, La], [N], [N] -> [N, 2 d + 2 c].  keep: dict with optional 'title_attn',
    'abst_attn' ([N, h, L, L]) and 'feat' ([N, F]) keep masks."""
    keep = keep or {}
    tv = text_vector(p, titles, n_heads, p_drop, keep.get("title_attn"))
    av = text_vector(p, absts, n_heads, p_drop, keep.get("abst_attn"))
    cv = F.embedding(categ, p["news_encoder.category_embedding.weight"], padding_idx=0)        # :107
    sv = F.embedding(subcateg, p["news_encoder.subcategory_embedding.weight"], padding_idx=0)  # :108
    return _keep(torch.cat([tv, av, cv, sv], -1), keep.get("feat"), p_drop)


def user_vector(p, hist, n_heads, p_drop=0.0, keep_attn=None):
    """UserEncoder.forward (:188-191): hist [B, H, F] (already normalised) -> [B, F]."""
    y = mhsa(p, "user_encoder.multi_head_self_attention.", hist, n_heads, p_drop, keep_attn)
    return additive(p, "user_encoder.additive_attention.", y)


def forward(p, batch, title_heads, user_heads, p_drop=0.0, keep=None, eps=1e-5, parts=False):
    """Model.forward (:216-257).  batch: dict of int64 / uint8 tensors.  keep (training with explicit masks): dict with
    'cand' / 'hist' sub-dicts for news_features and 'user_attn' [B, h, H, H]."""
    keep = keep or {}
    bt, ct = batch["browsed_titles"], batch["candidate_titles"]
    B, H, Lt = bt.shape
    C = ct.shape[1]
    La = batch["browsed_absts"].shape[2]
    cand = news_features(p, ct.reshape(B * C, Lt), batch["candidate_absts"].reshape(B * C, La),
                         batch["candidate_categ_ids"].reshape(-1), batch["candidate_subcateg_ids"].reshape(-1),
                         title_heads, p_drop, keep.get("cand")).view(B, C, -1)
    hist = news_features(p, bt.reshape(B * H, Lt), batch["browsed_absts"].reshape(B * H, La),
                         batch["browsed_categ_ids"].reshape(-1), batch["browsed_subcateg_ids"].reshape(-1),
                         title_heads, p_drop, keep.get("hist")).view(B, H, -1)
    normed = F.layer_norm(hist, (hist.shape[-1],), p["norm.weight"], p["norm.bias"], eps)
    user = user_vector(p, normed, user_heads, p_drop, keep.get("user_attn"))
    pred = torch.sum(user.unsqueeze(1) * cand, 2)
    pred = pred.masked_fill(batch["candidate_mask"] == 0, -1e9)
    if parts:
        return pred, dict(cand=cand, hist=hist, normed=normed, user=user)
    return pred


def loss_and_grads(params_np, batch_np, title_heads, user_heads, dtype=torch.float32, p_drop=0.0, keep=None):
    p = to_torch(params_np, dtype, requires_grad=True)
    batch = {k: torch.as_tensor(np.asarray(v)) for k, v in batch_np.items()}
    scores = forward(p, batch, title_heads, user_heads, p_drop, keep)
    loss = loss_fn(scores)
    loss.backward()
    grads = {k: (v.grad.detach().numpy() if v.grad is not None else np.zeros(v.shape, np.float32)) for k, v in p.items()}
    return scores.detach().numpy(), float(loss.detach()), grads
