"""Deterministic synthetic NRMS parameters and MIND-shaped batches.

There is no MIND / GloVe data offline, so tests, the golden-vector generator and
bench.py all draw their inputs from here.  Everything is produced by numpy's
``default_rng`` (PCG64, platform-stable), so a fixture only has to store the
*outputs* for a given (shape, seed); inputs and weights are regenerated.

Parameter names and shapes follow the reference model so state dicts interchange
(/root/reference/MIND_2020/model/nrms_v0.py:35-37,91-93,134-139,149-152,183-186).
Batch keys / dtypes / padding follow ``MyDataset.__getitem__``
(/root/reference/MIND_2020/data_handler.py:185-250): right-zero-padded titles,
left-aligned history, uint8 masks, int64 ids.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

import numpy as np

ENCODERS = ("news_encoder", "user_encoder")


@dataclass(frozen=True)
class Shape:
    """Problem shape. Names follow the reference's config fields (config.py:30-35,46,52,71,88)."""
    n_words: int = 45800          # V
    word_embed_size: int = 300    # d
    num_attention_heads: int = 10 # h
    query_vector_dim: int = 200   # q
    batch_size: int = 512         # B
    history_len: int = 50         # H
    n_candidates: int = 5         # C  (= sample_size + 1 in training)
    n_words_title: int = 30       # L

    def as_dict(self):
        return asdict(self)


# The fixture shapes of SURVEY.md section 8c.
G1_ODD = Shape(n_words=97, word_embed_size=60, num_attention_heads=10, query_vector_dim=32,
               batch_size=3, history_len=7, n_candidates=3, n_words_title=9)
G2_MIND = Shape(n_words=2000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                batch_size=4, history_len=50, n_candidates=5, n_words_title=30)
BENCH = Shape()


def param_names(with_output_proj: bool = False):
    """The 19 nrms_v0 parameter names, in the flat-buffer order used by the HIP path."""
    names = ["news_encoder.word_embedding.0.weight"]
    for enc in ENCODERS:
        a = enc + ".multihead_self_attention."
        names += [a + "W_Q.weight", a + "W_K.weight", a + "W_V.weight",
                  a + "W_Q.bias", a + "W_K.bias", a + "W_V.bias"]
        if with_output_proj:
            names += [a + "W_O.weight", a + "W_O.bias"]
        b = enc + ".additive_attention."
        names += [b + "linear.weight", b + "linear.bias", b + "attention_query_vector"]
    return names


def param_shapes(shape: Shape, with_output_proj: bool = False):
    V, d, q = shape.n_words, shape.word_embed_size, shape.query_vector_dim
    out = {"news_encoder.word_embedding.0.weight": (V, d)}
    for enc in ENCODERS:
        a = enc + ".multihead_self_attention."
        for w in ("W_Q", "W_K", "W_V"):
            out[a + w + ".weight"] = (d, d)
            out[a + w + ".bias"] = (d,)
        if with_output_proj:
            out[a + "W_O.weight"] = (d, d)
            out[a + "W_O.bias"] = (d,)
        b = enc + ".additive_attention."
        out[b + "linear.weight"] = (q, d)
        out[b + "linear.bias"] = (q,)
        out[b + "attention_query_vector"] = (q,)
    return {n: out[n] for n in param_names(with_output_proj)}


def make_params(shape: Shape, seed: int = 0, pad_row_zero: bool = True,
                with_output_proj: bool = False, weight_gain: float = 1.0):
    """float32 parameters with the reference's init distributions.

    Table ~ N(0, 0.4^2) (GloVe-like scale); Linear weights xavier-uniform
    (nrms_v0.py:41-44); biases U(+-1/sqrt(fan_in)) (torch Linear default);
    query vector U(-0.1, 0.1) (nrms_v0.py:92-93).
    ``pad_row_zero=False`` leaves a non-zero row 0 to exercise the
    "row 0 is used as stored but never receives gradient" semantics of
    ``padding_idx=0`` with ``from_pretrained`` (nrms_v0.py:134-136).
    """
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in param_shapes(shape, with_output_proj).items():
        if name.endswith("word_embedding.0.weight"):
            t = rng.normal(0.0, 0.4, size=shp)
            if pad_row_zero:
                t[0] = 0.0
        elif name.endswith("attention_query_vector"):
            t = rng.uniform(-0.1, 0.1, size=shp)
        elif name.endswith(".weight"):
            fan_out, fan_in = shp
            bound = weight_gain * np.sqrt(6.0 / (fan_in + fan_out))
            t = rng.uniform(-bound, bound, size=shp)
        else:  # Linear bias
            fan_in = shape.word_embed_size
            bound = 1.0 / np.sqrt(fan_in)
            t = rng.uniform(-bound, bound, size=shp)
        out[name] = np.ascontiguousarray(t, dtype=np.float32)
    return out


def make_batch(shape: Shape, seed: int = 1, ragged: bool = True, min_title: int = 5,
               empty_history_user: bool = False, all_pad_title: bool = False,
               mask_some_candidates: bool = False, batch_size: int | None = None):
    """A MIND-shaped batch dict of numpy arrays (keys and dtypes of data_handler.py:236-250).

    ragged: per-title length in [min_title, L] with the tail zeroed, per-user
    history length in [1, H] with trailing slots all zero (left-aligned history,
    data_handler.py:206-215).  Only the three keys the NRMS path reads
    (nrms_v0.py:248,250,272) plus ``browsed_mask``/``browsed_lens`` are produced.
    """
    B = shape.batch_size if batch_size is None else batch_size
    H, C, L, V = shape.history_len, shape.n_candidates, shape.n_words_title, shape.n_words
    rng = np.random.default_rng(seed)
    bt = rng.integers(1, V, size=(B, H, L), dtype=np.int64)
    ct = rng.integers(1, V, size=(B, C, L), dtype=np.int64)
    hist_len = np.full((B,), H, dtype=np.int64)
    if ragged:
        lo = min(min_title, L)
        tl_b = rng.integers(lo, L + 1, size=(B, H))
        tl_c = rng.integers(lo, L + 1, size=(B, C))
        pos = np.arange(L)
        bt = np.where(pos[None, None, :] < tl_b[..., None], bt, 0)
        ct = np.where(pos[None, None, :] < tl_c[..., None], ct, 0)
        hist_len = rng.integers(max(1, min(5, H)), H + 1, size=(B,))
    if empty_history_user and B > 1:
        hist_len[1] = 0
    slot = np.arange(H)
    bmask = (slot[None, :] < hist_len[:, None])
    bt = np.where(bmask[..., None], bt, 0)
    if all_pad_title:
        bt[0, 0, :] = 0          # a real (unmasked) history slot whose title is all padding
        ct[-1, -1, :] = 0
    cmask = np.ones((B, C), dtype=np.uint8)
    if mask_some_candidates:
        cmask[0, C - 1] = 0
        if B > 2:
            cmask[2, 1:] = 0     # a user with only the positive left
    return {
        "browsed_lens": hist_len.astype(np.int64),
        "browsed_titles": np.ascontiguousarray(bt, dtype=np.int64),
        "browsed_mask": bmask.astype(np.uint8),
        "candidate_titles": np.ascontiguousarray(ct, dtype=np.int64),
        "candidate_mask": cmask,
    }


def make_eval_impressions(n_imp: int, max_cand: int, seed: int = 7, min_cand: int = 2):
    """Padded impressions for the AUC path (train_eval.py:219-273): scores [n_imp, max_cand]
    with padded slots at -1e9 (nrms_v0.py:274) and ragged 0/1 label lists, each with at
    least one positive and one negative so roc_auc_score is defined."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_cand, max_cand + 1, size=(n_imp,))
    scores = np.full((n_imp, max_cand), -1e9, dtype=np.float32)
    labels = []
    for i, n in enumerate(lens):
        s = rng.normal(0, 0.06, size=(n,)).astype(np.float32)
        if n >= 6:                       # inject ties: the rank statistic must average them
            s[3] = s[1]
            s[5] = s[1]
        y = (rng.random(n) < 0.2).astype(np.int64)
        y[rng.integers(0, n)] = 1
        zeros = np.flatnonzero(y == 0)
        if zeros.size == 0:
            y[(int(np.argmax(y)) + 1) % n] = 0
        scores[i, :n] = s
        labels.append(y)
    return scores, labels


def dataset_fixture_inputs():
    """Hand-written samples for the MyDataset fixture (tests/golden/g6_dataset.npz): the positional sample
    format of data_handler.py:206-231, news indices 1-based, ``id2title_dict[news - 1]`` = padded word ids."""
    cfg = dict(history_len=5, n_words_title=4, n_words_abst=3, sample_size=2, max_candidate_size=6, batch_size=2)
    titles = {0: [11, 12, 0, 0], 1: [21, 0, 0, 0], 2: [31, 32, 33, 34], 3: [41, 42, 43, 0], 4: [51, 52, 0, 0],
              5: [61, 0, 0, 0]}
    absts = {i: [100 + i, 200 + i, 0] for i in range(6)}
    train = [
        # history, categ, subcateg, impressions (positive first), imp categ, imp subcateg
        [[3, 1], [2, 1], [5, 4], [2, 4, 5], [1, 2, 3], [7, 8, 9]],
        [[6, 5, 4, 3, 2], [1, 1, 2, 2, 3], [4, 4, 5, 5, 6], [1, 6, 3, 2, 5], [3, 2, 1], [9, 8, 7]],   # imps cut to sample_size+1
    ]
    evals = [
        [[2], [3], [6], [1, 2, 3, 4], [1, 1, 2, 2], [4, 5, 6, 7]],
        [[1, 2, 3, 4, 5], [1, 2, 3, 1, 2], [4, 5, 6, 4, 5], [6, 5, 4, 3, 2, 1], [3, 3, 2, 2, 1, 1], [9, 9, 8, 8, 7, 7]],
    ]
    return dict(config=cfg, id2title_dict=titles, id2abst_dict=absts, train_samples=train, eval_samples=evals)


def make_params_v1(shape: Shape, seed: int = 0):
    """Parameters with nrms_v1's names (model/nrms_v1.py:54-55,87,115): linear_layers.{0,1,2},
    output_linear (W_O), query_vector; same initial distributions as make_params."""
    from .engine import ModelDims, FlatLayout      # local import: engine imports this module
    dims = ModelDims(shape.n_words, shape.word_embed_size, shape.num_attention_heads, shape.query_vector_dim,
                     output_proj=True, style="v1")
    rng = np.random.default_rng(seed)
    out = {}
    for name, (_, shp, _) in FlatLayout(dims).entries.items():
        if name.endswith("word_embedding.weight"):
            t = rng.normal(0.0, 0.4, size=shp)
            t[0] = 0.0
        elif name.endswith("query_vector"):
            t = rng.uniform(-0.1, 0.1, size=shp)
        elif name.endswith(".weight"):
            bound = np.sqrt(6.0 / (shp[0] + shp[1]))
            t = rng.uniform(-bound, bound, size=shp)
        else:
            t = rng.uniform(-1.0, 1.0, size=shp) / np.sqrt(shape.word_embed_size)
        out[name] = np.ascontiguousarray(t, dtype=np.float32)
    return out


# ---------------------------------------------------------------------------------------------------------------
# nrms_naml (SURVEY f-3; model/nrms_naml.py:103-257): title + abstract through ONE word-level encoder (W_O, dropout on
# the attention probabilities), category / sub-category embeddings, LayerNorm on the history, a wide user encoder.
@dataclass(frozen=True)
class NamlShape:
    """Names follow config.py:30-31,45-49,68-72,77,87."""
    n_words: int = 45800
    word_embed_size: int = 300
    title_heads_num: int = 6
    query_vector_dim: int = 200
    category_nums: int = 19
    subcategory_nums: int = 294
    cate_embed_size: int = 100
    user_heads_num: int = 8
    query_vector_dim_large: int = 400
    batch_size: int = 512
    history_len: int = 50
    n_candidates: int = 5
    n_words_title: int = 20
    n_words_abst: int = 40

    @property
    def news_feature_size(self):
        return 2 * self.word_embed_size + 2 * self.cate_embed_size


G7_ODD = NamlShape(n_words=83, word_embed_size=48, title_heads_num=6, query_vector_dim=20, category_nums=5,
                   subcategory_nums=9, cate_embed_size=16, user_heads_num=8, query_vector_dim_large=36,
                   batch_size=3, history_len=5, n_candidates=3, n_words_title=7, n_words_abst=11)
G7_MIND = NamlShape(n_words=500, batch_size=2, history_len=50, n_candidates=5)


def naml_param_shapes(s: NamlShape):
    """The 27 tensors of nrms_naml.Model.state_dict(), in registration order (nrms_naml.py:106-117,181-186,200-207)."""
    d, F, q, Q = s.word_embed_size, s.news_feature_size, s.query_vector_dim, s.query_vector_dim_large
    out = {"news_encoder.category_embedding.weight": (s.category_nums, s.cate_embed_size),
           "news_encoder.subcategory_embedding.weight": (s.subcategory_nums, s.cate_embed_size),
           "news_encoder.word_embedding.weight": (s.n_words, d)}
    for enc, dm, qq in (("news_encoder", d, q), ("user_encoder", F, Q)):
        a = enc + ".multi_head_self_attention."
        for i in range(3):
            out[a + "linear_layers.%d.weight" % i] = (dm, dm)
            out[a + "linear_layers.%d.bias" % i] = (dm,)
        out[a + "output_linear.weight"] = (dm, dm)
        out[a + "output_linear.bias"] = (dm,)
        out[enc + ".additive_attention.query_vector"] = (qq,)        # a module's own parameters precede its children's
        out[enc + ".additive_attention.linear.weight"] = (qq, dm)
        out[enc + ".additive_attention.linear.bias"] = (qq,)
    out["norm.weight"] = (F,)
    out["norm.bias"] = (F,)
    return out


def make_params_naml(s: NamlShape, seed: int = 0):
    """Initial distributions of the reference's modules (torch defaults: Linear kaiming-uniform(a=sqrt 5) = U(+-1/sqrt
    fan_in), Embedding N(0,1) with a zero padding row; query vector U(-0.1, 0.1), nrms_naml.py:82); LayerNorm gets a
    random affine instead of (1, 0) so that its gradients are exercised."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in naml_param_shapes(s).items():
        if name.endswith("word_embedding.weight"):
            t = rng.normal(0.0, 0.4, size=shp)
            t[0] = 0.0
        elif name.endswith("category_embedding.weight"):
            t = rng.normal(0.0, 1.0, size=shp)
            t[0] = 0.0
        elif name.endswith("query_vector"):
            t = rng.uniform(-0.1, 0.1, size=shp)
        elif name == "norm.weight":
            t = rng.uniform(0.5, 1.5, size=shp)
        elif name == "norm.bias":
            t = rng.uniform(-0.2, 0.2, size=shp)
        elif name.endswith(".weight"):
            bound = 1.0 / np.sqrt(shp[1])
            t = rng.uniform(-bound, bound, size=shp)
        else:
            fan_in = s.news_feature_size if name.startswith("user_encoder") else s.word_embed_size
            t = rng.uniform(-1.0, 1.0, size=shp) / np.sqrt(fan_in)
        out[name] = np.ascontiguousarray(t, dtype=np.float32)
    return out


def make_batch_naml(s: NamlShape, seed: int = 1, batch_size: int | None = None, mask_some_candidates: bool = True):
    """The batch-dict keys nrms_naml.Model.forward reads (nrms_naml.py:217-228,245), shapes and dtypes of
    data_handler.py:236-250: ragged titles / abstracts, left-aligned histories, padding slots with category 0."""
    B = s.batch_size if batch_size is None else batch_size
    H, C = s.history_len, s.n_candidates
    rng = np.random.default_rng(seed)

    def texts(n_slots, L):
        ids = rng.integers(1, s.n_words, size=(B, n_slots, L), dtype=np.int64)
        ln = rng.integers(1, L + 1, size=(B, n_slots))
        return np.where(np.arange(L)[None, None, :] < ln[..., None], ids, 0)

    bt, ba = texts(H, s.n_words_title), texts(H, s.n_words_abst)
    ct, ca = texts(C, s.n_words_title), texts(C, s.n_words_abst)
    bc = rng.integers(1, s.category_nums, size=(B, H), dtype=np.int64)
    bs = rng.integers(1, s.subcategory_nums, size=(B, H), dtype=np.int64)
    cc = rng.integers(1, s.category_nums, size=(B, C), dtype=np.int64)
    cs = rng.integers(1, s.subcategory_nums, size=(B, C), dtype=np.int64)
    hist_len = rng.integers(1, H + 1, size=(B,))
    if B > 1:
        hist_len[1] = 0
    live = np.arange(H)[None, :] < hist_len[:, None]
    bt, ba = np.where(live[..., None], bt, 0), np.where(live[..., None], ba, 0)
    bc, bs = np.where(live, bc, 0), np.where(live, bs, 0)
    ba[0, 0, :] = 0                                     # a news item without an abstract
    cmask = np.ones((B, C), dtype=np.uint8)
    if mask_some_candidates:
        cmask[0, C - 1] = 0
    c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    return {"browsed_titles": c64(bt), "browsed_absts": c64(ba), "browsed_categ_ids": c64(bc),
            "browsed_subcateg_ids": c64(bs), "candidate_titles": c64(ct), "candidate_absts": c64(ca),
            "candidate_categ_ids": c64(cc), "candidate_subcateg_ids": c64(cs), "candidate_mask": cmask,
            "browsed_lens": hist_len.astype(np.int64), "browsed_mask": live.astype(np.uint8)}


# ---- HieRec-style hierarchical interest model (model/hierec_hip.py; BASELINE configs[3]; parity unpinned) ----------------------
def make_params_hierec(shape: Shape, n_sub: int = 294, n_top: int = 19, seed: int = 0):
    """The NRMS news encoder's parameters (make_params) + sub-topic / topic embedding tables (N(0, 0.1^2), zero padding row)
    + the three levels' additive attentions (xavier-uniform, U(+-1/sqrt fan_in), U(-0.1, 0.1))."""
    base = make_params(shape, seed=seed)
    out = {k: v for k, v in base.items() if k.startswith("news_encoder.")}
    rng = np.random.default_rng(seed + 1000)
    d, q = shape.word_embed_size, shape.query_vector_dim
    for name, n in (("subtopic_embedding.weight", n_sub), ("topic_embedding.weight", n_top)):
        t = rng.normal(0.0, 0.1, size=(n, d))
        t[0] = 0.0
        out[name] = t.astype(np.float32)
    for lv in ("subtopic_attention", "topic_attention", "user_attention"):
        a = np.sqrt(6.0 / (d + q))
        out[lv + ".linear.weight"] = rng.uniform(-a, a, size=(q, d)).astype(np.float32)
        out[lv + ".linear.bias"] = rng.uniform(-1, 1, size=(q,)).astype(np.float32) / np.float32(np.sqrt(d))
        out[lv + ".attention_query_vector"] = rng.uniform(-0.1, 0.1, size=(q,)).astype(np.float32)
    return out


def make_batch_hierec(shape: Shape, n_sub: int = 294, n_top: int = 19, seed: int = 1, batch_size: int | None = None,
                      empty_history_user: bool = False, mask_some_candidates: bool = False, consistent_topics: bool = True,
                      zipf: float = 1.2):
    """make_batch + the category keys of data_handler.py:236-250.  Sub-topics are drawn with a Zipf-like skew (a few sub-topics
    hold most clicks, as in MIND) and, with consistent_topics, each sub-topic belongs to one topic; padding slots carry id 0."""
    b = make_batch(shape, seed=seed, batch_size=batch_size, empty_history_user=empty_history_user,
                   mask_some_candidates=mask_some_candidates)
    rng = np.random.default_rng(seed + 77)
    B, H = b["browsed_titles"].shape[:2]
    Cn = b["candidate_titles"].shape[1]
    w = 1.0 / np.arange(1, n_sub) ** zipf
    w /= w.sum()
    owner = rng.integers(1, n_top, size=n_sub)

    def draw(shape_):
        s = rng.choice(np.arange(1, n_sub), size=shape_, p=w)
        t = owner[s] if consistent_topics else rng.integers(1, n_top, size=shape_)
        return s.astype(np.int64), t.astype(np.int64)

    bs, btp = draw((B, H))
    cs, ctp = draw((B, Cn))
    valid = np.asarray(b["browsed_mask"]).astype(bool)
    b["browsed_subcateg_ids"] = np.where(valid, bs, 0)
    b["browsed_categ_ids"] = np.where(valid, btp, 0)
    b["candidate_subcateg_ids"], b["candidate_categ_ids"] = cs, ctp
    return b


# ---- user-news graph encoder (model/graph_hip.py; BASELINE configs[4]; parity unpinned) ---------------------------------------
def make_params_graph(shape: Shape, seed: int = 0):
    """The NRMS news encoder's parameters (make_params) + the two aggregators' additive attentions."""
    base = make_params(shape, seed=seed)
    out = {k: v for k, v in base.items() if k.startswith("news_encoder.")}
    rng = np.random.default_rng(seed + 2000)
    d, q = shape.word_embed_size, shape.query_vector_dim
    for lv in ("neighbor_attention", "user_attention"):
        a = np.sqrt(6.0 / (d + q))
        out[lv + ".linear.weight"] = rng.uniform(-a, a, size=(q, d)).astype(np.float32)
        out[lv + ".linear.bias"] = rng.uniform(-1, 1, size=(q,)).astype(np.float32) / np.float32(np.sqrt(d))
        out[lv + ".attention_query_vector"] = rng.uniform(-0.1, 0.1, size=(q,)).astype(np.float32)
    return out


def make_batch_graph(shape: Shape, n_neighbors: int = 8, seed: int = 1, batch_size: int | None = None, empty_history_user: bool = False,
                     mask_some_candidates: bool = False, zipf: float = 1.1):
    """make_batch + ``neighbor_rows`` [B * (H + C), K]: a sampled sub-graph induced on the batch's own news slots.  A slot has
    0 .. K neighbours (a prefix of its row, -1 after it), drawn with a Zipf-like skew over the slots (a few popular news are
    everybody's neighbour, as in a click graph); one row carries an out-of-range entry in the middle of its list."""
    b = make_batch(shape, seed=seed, batch_size=batch_size, empty_history_user=empty_history_user, mask_some_candidates=mask_some_candidates)
    rng = np.random.default_rng(seed + 99)
    B, H = b["browsed_titles"].shape[:2]
    N = B * (H + b["candidate_titles"].shape[1])
    w = 1.0 / np.arange(1, N + 1) ** zipf
    w /= w.sum()
    popular = rng.permutation(N)
    nbr = popular[rng.choice(N, size=(N, n_neighbors), p=w)].astype(np.int64)
    cnt = rng.integers(0, n_neighbors + 1, size=(N,))
    nbr = np.where(np.arange(n_neighbors)[None, :] < cnt[:, None], nbr, -1)
    if N > 3 and n_neighbors > 2:
        nbr[3, :] = [5 % N, -1] + [7 % N] * (n_neighbors - 2)
    b["neighbor_rows"] = nbr
    return b
