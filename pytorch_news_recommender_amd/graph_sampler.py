"""Neighbour sampling for the user-news graph encoder (model/graph_hip.py; SURVEY section 8 row f-4, parity unpinned): the
``neighbor_rows`` key of its batch dict, drawn from the click graph INDUCED on a batch -- host-side index preparation (numpy), the
counterpart of what ``data_handler.MyDataset`` does for the other keys.  The reference has no sampler (it has no graph model).

A news slot's neighbours are news co-clicked with it: slot r shows news j; a user u of the batch who clicked j is drawn, then one
of u's other clicks -- as a ROW of the batch's slot numbering (history slot (b, k) = b * H + k; candidate (b, c) = B * H + b * C + c).
News identity = equal title rows (a batch dict carries word ids, not news ids).  -1 = no neighbour (nobody in the batch clicked
the slot's news, or the draw hit the same news)."""
from __future__ import annotations

import numpy as np


def induced_neighbor_rows(browsed_titles, browsed_mask, candidate_titles, n_neighbors=8, seed=0):
    bt, ct = np.asarray(browsed_titles), np.asarray(candidate_titles)
    valid = np.asarray(browsed_mask).astype(bool)
    B, H, L = bt.shape
    C = ct.shape[1]
    N = B * (H + C)
    rng = np.random.default_rng(seed)
    titles = np.concatenate([bt.reshape(B * H, L), ct.reshape(B * C, L)], 0)
    _, news = np.unique(titles, axis=0, return_inverse=True)                 # news identity of every slot
    news = news.reshape(-1)
    clicks = np.flatnonzero(valid.reshape(-1))                               # history rows that are real clicks
    order = np.argsort(news[clicks], kind="stable")
    by_news = clicks[order]                                                  # click rows grouped by the news they show
    n_news = int(news.max()) + 1 if N else 0
    cnt = np.bincount(news[clicks], minlength=n_news)
    start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    hist_len = valid.sum(1)
    out = -np.ones((N, n_neighbors), dtype=np.int64)
    if N == 0 or clicks.size == 0:
        return out
    c_r = cnt[news]                                                          # clickers of each slot's news
    has = c_r > 0
    pick = rng.random((N, n_neighbors))
    click_row = by_news[np.minimum(start[news][:, None] + (pick * c_r[:, None]).astype(np.int64), len(by_news) - 1)]
    user = click_row // H                                                    # a user who clicked this slot's news ...
    slot = (rng.random((N, n_neighbors)) * hist_len[user]).astype(np.int64)  # ... and one of that user's clicks (left-aligned history)
    nb = user * H + np.minimum(slot, H - 1)
    ok = has[:, None] & valid.reshape(-1)[nb] & (news[nb] != news[:, None])
    out[ok] = nb[ok]
    return out
