"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (PyTorch CPU tensors, fp32 or fp64, autograd for gradients) of the
NRMS train/eval hot path of 0215Arthur/Pytorch_News_Recommender.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline -- the product package
(``pytorch_news_recommender_amd``) never imports it and fails loudly without its
HIP library.

Parity pin: every function below is checked against outputs of the *imported
reference itself* (run in the build container by ``tests/golden/gen_golden.py``)
through the fixtures committed under ``tests/golden/*.npz``
(``tests/test_oracle_golden.py``).  The reference has no tests or golden vectors
of its own, so those fixtures are the pin.

Each function cites the reference lines it restates (paths relative to
/root/reference/MIND_2020/).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

ENCODERS = ("news_encoder", "user_encoder")


def to_torch(params: dict, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in params.items():
        t = torch.as_tensor(np.asarray(v)).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def scaled_dot_product_attention(Q, K, V, pair_mask=None):
    """model/nrms_v0.py:13-23 -- softmax(Q K^T / sqrt(d_k)) V.  v0 ignores any mask
    (mask code commented out at :15-19), so pad tokens are ordinary keys.
    ``pair_mask`` reproduces v1's ``masked_fill(_mask == 0, -1e9)`` (model/nrms_v1.py:27-33)."""
    d_k = Q.shape[-1]
    scores = torch.matmul(Q, K.transpose(-1, -2)) / math.sqrt(d_k)
    if pair_mask is not None:
        scores = scores.masked_fill(pair_mask == 0, -1e9)
    attn = F.softmax(scores, dim=-1)
    return torch.matmul(attn, V)


def multihead_self_attention(p, prefix, X, n_heads, mask=None):
    """model/nrms_v0.py:46-76 (three Linear(d,d) with bias, split into heads, attention,
    heads concatenated, NO output projection).  When ``prefix+'W_O.weight'`` is present
    the v1 ``output_linear`` is applied (model/nrms_v1.py:55,80); ``mask`` [N,S] builds
    v1's pairwise mask ``mask[:,None,:]*mask[:,:,None]`` broadcast over heads
    (model/nrms_v1.py:29-32)."""
    N, S, d = X.shape
    d_k = d // n_heads

    def proj(w):
        y = F.linear(X, p[prefix + w + ".weight"], p[prefix + w + ".bias"])
        return y.view(N, S, n_heads, d_k).transpose(1, 2)

    q, k, v = proj("W_Q"), proj("W_K"), proj("W_V")
    pair = None
    if mask is not None:
        m = mask.to(X.dtype)
        pair = (m.unsqueeze(1) * m.unsqueeze(2)).unsqueeze(1)
    ctx = scaled_dot_product_attention(q, k, v, pair)
    ctx = ctx.transpose(1, 2).contiguous().view(N, S, n_heads * d_k)
    if prefix + "W_O.weight" in p:
        ctx = F.linear(ctx, p[prefix + "W_O.weight"], p[prefix + "W_O.bias"])
    return ctx


def additive_attention(p, prefix, X, mask=None):
    """model/nrms_v0.py:100-126 -- temp = tanh(Linear(X)); w = softmax(temp . q, dim=1);
    out = w^T X.  ``mask`` [N,S] reproduces v1's masked_fill(mask==0,-1e9) before the
    softmax (model/nrms_v1.py:100-101)."""
    temp = torch.tanh(F.linear(X, p[prefix + "linear.weight"], p[prefix + "linear.bias"]))
    s = torch.matmul(temp, p[prefix + "attention_query_vector"])
    if mask is not None:
        s = s.masked_fill(mask == 0, -1e9)
    w = F.softmax(s, dim=1)
    return torch.bmm(w.unsqueeze(1), X).squeeze(1)


def _apply_keep(x, keep, p_drop):
    """Dropout with an explicit keep mask: y = x * keep / (1 - p) (torch.nn.Dropout
    semantics, model/nrms_v0.py:137,171-173)."""
    if keep is None or p_drop == 0.0:
        return x
    return x * keep.to(x.dtype) / (1.0 - p_drop)


def v1_to_v0_names(params):
    """nrms_v1's parameter names (model/nrms_v1.py:54-55,87,115) -> the v0-style keys the functions
    here use; the math is shared, only the names and the presence of W_O differ."""
    out = {}
    for k, v in params.items():
        k = k.replace("multi_head_self_attention.linear_layers.0", "multihead_self_attention.W_Q")
        k = k.replace("multi_head_self_attention.linear_layers.1", "multihead_self_attention.W_K")
        k = k.replace("multi_head_self_attention.linear_layers.2", "multihead_self_attention.W_V")
        k = k.replace("multi_head_self_attention.output_linear", "multihead_self_attention.W_O")
        k = k.replace("additive_attention.query_vector", "additive_attention.attention_query_vector")
        if k == "news_encoder.word_embedding.weight":
            k = "news_encoder.word_embedding.0.weight"
        out[k] = v
    return out


def news_encoder(p, ids, n_heads, p_drop=0.0, keep_embed=None, keep_ctx=None, training=False,
                 generator=None, embed_dropout=True, mask=None, mask_mode=0):
    """model/nrms_v0.py:154-176 -- embedding (row 0 = pad row, used as stored) -> dropout ->
    MHSA -> F.dropout -> additive attention.  ids [N,L] int64 -> [N,d].

    training=True with keep masks None draws torch dropout masks (CPU-baseline mode);
    explicit keep masks let tests replay the HIP path's counter-based masks."""
    table = p["news_encoder.word_embedding.0.weight"]
    X = F.embedding(ids, table, padding_idx=0)
    if embed_dropout:                       # nrms_v1 has no embedding dropout (nrms_v1.py:159-161)
        if training and keep_embed is None and p_drop > 0:
            X = F.dropout(X, p_drop, True)
        else:
            X = _apply_keep(X, keep_embed, p_drop)
    ctx = multihead_self_attention(p, "news_encoder.multihead_self_attention.", X, n_heads,
                                   mask=mask if (mask_mode & 1) else None)
    if training and keep_ctx is None and p_drop > 0:
        ctx = F.dropout(ctx, p_drop, True)
    else:
        ctx = _apply_keep(ctx, keep_ctx, p_drop)
    return additive_attention(p, "news_encoder.additive_attention.", ctx, mask if (mask_mode & 2) else None)


def user_encoder(p, news_vectors, n_heads, mask=None, mask_mode=0):
    """model/nrms_v0.py:188-199 -- MHSA over the clicked-news vectors + additive pooling;
    no dropout, no mask (v1's UserEncoder.forward accepts attn_masks, nrms_v1.py:208-211)."""
    ctx = multihead_self_attention(p, "user_encoder.multihead_self_attention.", news_vectors, n_heads,
                                   mask=mask if (mask_mode & 1) else None)
    return additive_attention(p, "user_encoder.additive_attention.", ctx, mask if (mask_mode & 2) else None)


def click_scores(cand_vec, user_vec, cand_mask=None):
    """model/nrms_v0.py:205-216 (bmm logits, no sigmoid) + :272-274 (masked_fill -1e9)."""
    s = torch.bmm(cand_vec, user_vec.unsqueeze(-1)).squeeze(-1)
    if cand_mask is not None:
        s = s.masked_fill(cand_mask == 0, -1e9)
    return s


def forward(p, batch, n_heads, p_drop=0.0, keep=None, training=False, per_slot=False, news_heads=None,
            embed_dropout=True):
    """model/nrms_v0.py:230-276 -- scores [B,C].

    per_slot=False encodes all B*(H+C) titles in one batched call (same math);
    per_slot=True is the "reference-shaped" mode: one encoder call per slot in a Python
    loop + torch.stack (nrms_v0.py:255-260), which is what makes the reference's backward
    build one dense [V,d] embedding gradient per slot.  ``keep`` is an optional dict
    {'embed': [B*(H+C),L,d], 'ctx': same} of explicit dropout keep masks in the HIP
    path's title order (all history titles user-major, then all candidate titles)."""
    bt = torch.as_tensor(batch["browsed_titles"]).long()
    ct = torch.as_tensor(batch["candidate_titles"]).long()
    B, H, L = bt.shape
    C = ct.shape[1]
    nh = n_heads if news_heads is None else news_heads      # nrms_v1: title_heads_num (nrms_v1.py:122)
    if per_slot:
        assert keep is None
        cand = torch.stack([news_encoder(p, x, nh, p_drop, training=training, embed_dropout=embed_dropout)
                            for x in ct.permute(1, 0, 2)], dim=1)
        hist = torch.stack([news_encoder(p, x, nh, p_drop, training=training, embed_dropout=embed_dropout)
                            for x in bt.permute(1, 0, 2)], dim=1)
    else:
        ids = torch.cat([bt.reshape(B * H, L), ct.reshape(B * C, L)], dim=0)
        ke = kc = None
        if keep is not None:
            ke, kc = keep.get("embed"), keep.get("ctx")
        nv = news_encoder(p, ids, nh, p_drop, ke, kc, training=training, embed_dropout=embed_dropout)
        hist = nv[:B * H].view(B, H, -1)
        cand = nv[B * H:].view(B, C, -1)
    user = user_encoder(p, hist, n_heads)
    cm = batch.get("candidate_mask")
    cm = None if cm is None else torch.as_tensor(cm)
    return click_scores(cand, user, cm), {"hist": hist, "cand": cand, "user": user}


def loss_fn(scores):
    """train_eval.py:63,116-117 -- CrossEntropyLoss with every label = 0 (the positive is
    always candidate slot 0), mean over the batch."""
    y = torch.zeros(scores.shape[0], dtype=torch.long)
    return F.cross_entropy(scores, y)


def loss_and_grads(params_np, batch, n_heads, dtype=torch.float32, p_drop=0.0, keep=None,
                   per_slot=False, news_heads=None, embed_dropout=True):
    """One forward + backward (train_eval.py:111-126).  Returns scores, loss, dict of grads
    (numpy).  The embedding gradient is dense [V,d] with row 0 zero (padding_idx=0)."""
    p = to_torch(params_np, dtype, requires_grad=True)
    scores, aux = forward(p, batch, n_heads, p_drop, keep, per_slot=per_slot, news_heads=news_heads,
                          embed_dropout=embed_dropout)
    loss = loss_fn(scores)
    loss.backward()
    grads = {k: (v.grad.detach().numpy().copy() if v.grad is not None
                 else np.zeros(tuple(v.shape), dtype=np.float64 if dtype == torch.float64 else np.float32))
             for k, v in p.items()}
    aux = {k: v.detach().numpy() for k, v in aux.items()}
    return scores.detach().numpy(), float(loss.detach()), grads, aux


# --------------------------------------------------------------------------------------
# optimizer (train_eval.py:48,127): torch.optim.Adam defaults, lr = config.learning_rate
# --------------------------------------------------------------------------------------
def adam_step(param, grad, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad), single tensor, in numpy:
    m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  ``step`` is 1-based."""
    m[:] = b1 * m + (1.0 - b1) * grad
    v[:] = b2 * v + (1.0 - b2) * grad * grad
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    param[:] = param - (lr / bc1) * (m / denom)


def train_steps(params_np, batches, n_heads, lr=1e-3, dtype=torch.float32):
    """``len(batches)`` iterations of train_eval.py:111-127 with dropout = 0: forward, CE,
    zero_grad, backward, Adam.  Returns the parameters after the last step and the losses."""
    npdt = np.float64 if dtype == torch.float64 else np.float32
    params = {k: np.asarray(v, dtype=npdt).copy() for k, v in params_np.items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v_ = {k: np.zeros_like(v) for k, v in params.items()}
    losses = []
    for t, batch in enumerate(batches, start=1):
        _, loss, grads, _ = loss_and_grads(params, batch, n_heads, dtype)
        losses.append(loss)
        for k in params:
            adam_step(params[k], grads[k].astype(npdt), m[k], v_[k], t, lr)
    return params, losses


class ReferenceShapedTrainer:
    """CPU baseline for bench.py (kind="port"): the reference's train step as it runs --
    per-slot encoder loop, dropout on (torch RNG), dense per-slot embedding gradients
    accumulated by autograd, stock torch.optim.Adam(lr=1e-3) over all parameters
    (model/nrms_v0.py:255-260 + train_eval.py:111-127)."""

    def __init__(self, params_np, n_heads, p_drop=0.2, lr=1e-3):
        self.p = to_torch(params_np, torch.float32, requires_grad=True)
        self.n_heads = n_heads
        self.p_drop = p_drop
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr)

    def step(self, batch):
        scores, _ = forward(self.p, batch, self.n_heads, self.p_drop, training=True, per_slot=True)
        self.opt.zero_grad()
        loss = loss_fn(scores)
        loss.backward()
        self.opt.step()
        return float(loss.detach())


# --------------------------------------------------------------------------------------
# evaluation (train_eval.py:219-273, evaluation.py:26-27)
# --------------------------------------------------------------------------------------
def roc_auc(y_true, y_score):
    """Rank-statistic AUC with tie averaging -- what sklearn.metrics.roc_auc_score
    (evaluation.py:26-27) computes for binary labels: (sum of positive mid-ranks -
    n_pos (n_pos+1)/2) / (n_pos n_neg), in float64."""
    y_true = np.asarray(y_true)
    y_score = np.asarray(y_score, dtype=np.float64)
    n = y_score.shape[0]
    order = np.argsort(y_score, kind="mergesort")
    s = y_score[order]
    ranks = np.empty(n, dtype=np.float64)
    i = 0
    while i < n:
        j = i
        while j + 1 < n and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    pos = y_true == 1
    n_pos = int(pos.sum())
    n_neg = n - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("AUC undefined: only one class present")
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def mean_impression_auc(scores, labels):
    """train_eval.py:219-227,255-271 -- per impression AUC on the un-padded prefix
    ``scores[i][:len(y_true)]``, unweighted mean over impressions."""
    aucs = [roc_auc(y, scores[i][:len(y)]) for i, y in enumerate(labels)]
    return float(np.mean(aucs)), aucs
