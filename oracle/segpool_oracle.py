"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (PyTorch CPU, fp32 / fp64, autograd for gradients) of the gather + additive-attention aggregate of
``include/nrms_hip.h`` (``nrms_segment_pool_fwd`` / ``_bwd``), the aggregation step of SURVEY section 8 row f-4 (BASELINE
configs 4-5: a HieRec-style hierarchical interest model, a user-news graph encoder).

PARITY UNPINNED: the reference holds no implementation of either model (``/root/reference/MIND_2020/model/tanr.py`` is an empty
file) and no fixture for this operation, so there is nothing of the reference's to check this restatement against.  What it
follows is the reference's own additive attention (``model/nrms_v0.py:100-126``: ``temp = tanh(linear(x))``,
``softmax(temp . query_vector)``, weighted sum), applied to the members of an index list instead of the rows of a fixed-length
sequence; the hierarchical model below composes it as HieRec (Qi et al., ACL 2021) describes its interest tree.
"""
from __future__ import annotations

import torch


def segment_pool(x, w_add, b_add, q_vec, seg_ptr, idx):
    """x [R, d]; seg_ptr [n_seg + 1]; idx [nnz] -> out [n_seg, d].  Plain loops: small cases only."""
    logit = torch.tanh(x @ w_add.t() + b_add) @ q_vec                     # nrms_v0.py:108-110, once per row
    out = []
    for s in range(len(seg_ptr) - 1):
        members = idx[int(seg_ptr[s]):int(seg_ptr[s + 1])]
        if len(members) == 0:
            out.append(torch.zeros(x.shape[1], dtype=x.dtype))
            continue
        m = torch.as_tensor(members, dtype=torch.long)
        alpha = torch.softmax(logit[m], dim=0)                             # nrms_v0.py:110-112 over the segment
        out.append((alpha.unsqueeze(1) * x[m]).sum(0))                     # nrms_v0.py:124-125
    return torch.stack(out) if out else torch.zeros(0, x.shape[1], dtype=x.dtype)


def hierarchical_interest(news_vec, valid, topic, subtopic, p):
    """One user's interest tree (HieRec's three levels) from its clicked-news vectors.
    news_vec [H, d], valid [H] bool, topic / subtopic [H] int.  p: dict with 'sub', 'top', 'user' -> (w_add, b_add, q_vec) and
    'E_sub' [n_sub, d], 'E_top' [n_top, d].  Returns (u_sub {subtopic id: vec}, n_sub {id: clicks}, u_top {topic id: vec},
    n_top {id: clicks}, u_user vec, n_clicks)."""
    d = news_vec.shape[1]
    slots = [k for k in range(news_vec.shape[0]) if bool(valid[k])]
    u_sub, n_sub, sub_topic = {}, {}, {}
    for s in sorted({int(subtopic[k]) for k in slots}):
        members = [k for k in slots if int(subtopic[k]) == s]
        r = segment_pool(news_vec, *p["sub"], [0, len(members)], members)[0]
        u_sub[s] = r + p["E_sub"][s]
        n_sub[s] = len(members)
        sub_topic[s] = int(topic[members[0]])
    u_top, n_top = {}, {}
    for t in sorted({sub_topic[s] for s in u_sub}):
        subs = [s for s in sorted(u_sub) if sub_topic[s] == t]
        rows = torch.stack([u_sub[s] for s in subs])
        r = segment_pool(rows, *p["top"], [0, len(subs)], list(range(len(subs))))[0]
        u_top[t] = r + p["E_top"][t]
        n_top[t] = sum(n_sub[s] for s in subs)
    if u_top:
        tops = sorted(u_top)
        rows = torch.stack([u_top[t] for t in tops])
        u_user = segment_pool(rows, *p["user"], [0, len(tops)], list(range(len(tops))))[0]
    else:
        u_user = torch.zeros(d, dtype=news_vec.dtype)
    return u_sub, n_sub, u_top, n_top, u_user, len(slots)


def hierarchical_scores(cand_vec, cand_topic, cand_subtopic, tree, lambda_sub=0.7, lambda_top=0.15):
    """HieRec's hierarchical matching for one user: cand_vec [C, d] -> [C].
    o = l_s f_s <n, u_sub[s_c]> + l_t f_t <n, u_top[t_c]> + (1 - l_s - l_t) <n, u_user>, f = share of the user's clicks in that
    sub-topic / topic (0 when the user never clicked there)."""
    u_sub, n_sub, u_top, n_top, u_user, n = tree
    out = []
    for c in range(cand_vec.shape[0]):
        s, t = int(cand_subtopic[c]), int(cand_topic[c])
        o = (1.0 - lambda_sub - lambda_top) * (cand_vec[c] @ u_user)
        if s in u_sub:
            o = o + lambda_sub * (n_sub[s] / n) * (cand_vec[c] @ u_sub[s])
        if t in u_top:
            o = o + lambda_top * (n_top[t] / n) * (cand_vec[c] @ u_top[t])
        out.append(o)
    return torch.stack(out)


def hierec_forward(p, batch, n_heads, lambda_sub=0.7, lambda_top=0.15):
    """The whole model of ``pytorch_news_recommender_amd/model/hierec_hip.py`` (its docstring is the specification): NRMS news
    encoder (oracle/nrms_oracle.py, model/nrms_v0.py:154-176) -> interest tree per user -> hierarchical scores [B, C], masked
    candidates at -1e9 (nrms_v0.py:272-274).  p: torch tensors under the model's parameter names."""
    from . import nrms_oracle as orc
    bt = torch.as_tensor(batch["browsed_titles"]).long()
    ct = torch.as_tensor(batch["candidate_titles"]).long()
    B, H, L = bt.shape
    C = ct.shape[1]
    nv = orc.news_encoder(p, torch.cat([bt.reshape(B * H, L), ct.reshape(B * C, L)], 0), n_heads)
    hist, cand = nv[:B * H].view(B, H, -1), nv[B * H:].view(B, C, -1)
    lv = lambda m: (p[m + ".linear.weight"], p[m + ".linear.bias"], p[m + ".attention_query_vector"])
    pp = {"sub": lv("subtopic_attention"), "top": lv("topic_attention"), "user": lv("user_attention"),
          "E_sub": p["subtopic_embedding.weight"], "E_top": p["topic_embedding.weight"]}
    valid = torch.as_tensor(batch["browsed_mask"]).bool()
    out = []
    for b in range(B):
        tree = hierarchical_interest(hist[b], valid[b], batch["browsed_categ_ids"][b], batch["browsed_subcateg_ids"][b], pp)
        if tree[5] == 0:
            out.append(torch.zeros(C, dtype=nv.dtype))         # no valid click: every interest vector is zero
            continue
        out.append(hierarchical_scores(cand[b], batch["candidate_categ_ids"][b], batch["candidate_subcateg_ids"][b], tree,
                                       lambda_sub, lambda_top))
    s = torch.stack(out)
    cm = batch.get("candidate_mask")
    if cm is not None:
        s = s.masked_fill(torch.as_tensor(cm) == 0, -1e9)
    return s


def graph_forward(p, batch, n_heads):
    """The whole model of ``pytorch_news_recommender_amd/model/graph_hip.py`` (its docstring is the specification): NRMS news
    encoder -> g_r = n_r + AddPool(neighbours of r) -> h_b = AddPool(clicked slots of b) -> <g_cand, h_b>, masked candidates at
    -1e9 (nrms_v0.py:205-216,272-274)."""
    from . import nrms_oracle as orc
    bt = torch.as_tensor(batch["browsed_titles"]).long()
    ct = torch.as_tensor(batch["candidate_titles"]).long()
    B, H, L = bt.shape
    C = ct.shape[1]
    N = B * (H + C)
    nv = orc.news_encoder(p, torch.cat([bt.reshape(B * H, L), ct.reshape(B * C, L)], 0), n_heads)
    lv = lambda m: (p[m + ".linear.weight"], p[m + ".linear.bias"], p[m + ".attention_query_vector"])
    nbr = batch["neighbor_rows"]
    ptr, idx = [0], []
    for r in range(N):
        idx += [int(v) for v in nbr[r] if 0 <= int(v) < N]
        ptr.append(len(idx))
    g = nv + segment_pool(nv, *lv("neighbor_attention"), ptr, idx)
    valid = batch["browsed_mask"]
    ptr, idx = [0], []
    for b in range(B):
        idx += [b * H + k for k in range(H) if valid[b][k]]
        ptr.append(len(idx))
    h = segment_pool(g, *lv("user_attention"), ptr, idx)
    cm = batch.get("candidate_mask")
    return orc.click_scores(g[B * H:].view(B, C, -1), h, None if cm is None else torch.as_tensor(cm))
