"""HieRec-style hierarchical interest model (model/hierec_hip.py; SURVEY section 8 row f-4, BASELINE configs[3]) against
oracle/segpool_oracle.py.  PARITY UNPINNED: the reference holds no implementation of this model (model/tanr.py is empty), so the
oracle restates the specification in the model's docstring on top of the reference's additive attention (nrms_v0.py:100-126);
what these tests pin is that the HIP path computes that specification -- index lists, scores, every gradient, one Adam step."""
import ctypes as C

import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import _lib, synth

pytestmark = pytest.mark.gpu

N_SUB, N_TOP = 23, 7


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_hierec(shape, params, precision="fp32", n_sub=N_SUB, n_top=N_TOP, device="cuda"):
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.model.hierec_hip import Model
    cfg = Config("hierec")
    cfg.__nrms__()
    cfg.word_embed_size, cfg.num_attention_heads, cfg.query_vector_dim = shape.word_embed_size, shape.num_attention_heads, shape.query_vector_dim
    cfg.dropout, cfg.precision = 0.0, precision
    cfg.subcategory_nums, cfg.category_nums = n_sub, n_top
    m = Model(cfg, pretrained_word_embedding=params["news_encoder.word_embedding.0.weight"])
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return m.to(device)


def tbatch(batch):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}


def python_tree(valid, topic, sub):
    """Groups of one user in order of first occurrence: [(sub id, topic id of its first click, member slots)], [(topic id, sub-group
    numbers)]."""
    subs, tops = [], []
    for k in range(len(valid)):
        if not valid[k]:
            continue
        for g in subs:
            if g[0] == sub[k]:
                g[2].append(k)
                break
        else:
            subs.append([int(sub[k]), int(topic[k]), [k]])
    for gi, g in enumerate(subs):
        for t in tops:
            if t[0] == g[1]:
                t[1].append(gi)
                break
        else:
            tops.append([g[1], [gi]])
    return subs, tops


@pytest.mark.parametrize("B,H,consistent", [(5, 50, True), (3, 64, False), (130, 33, True), (2, 1, True)])
def test_tree_lists_against_a_python_grouping(B, H, consistent):
    lib = _lib.load()
    shape = synth.Shape(n_words=50, word_embed_size=8, num_attention_heads=2, query_vector_dim=4, batch_size=B, history_len=H,
                        n_candidates=4, n_words_title=3)
    batch = synth.make_batch_hierec(shape, N_SUB, N_TOP, seed=B + H, empty_history_user=B > 1, consistent_topics=consistent)
    valid = torch.from_numpy(batch["browsed_mask"].astype(np.uint8)).cuda()
    topic = torch.from_numpy(batch["browsed_categ_ids"]).cuda()
    sub = torch.from_numpy(batch["browsed_subcateg_ids"]).cuda()
    n = B * H
    i32 = lambda m: torch.full((m,), -7, dtype=torch.int32, device="cuda")
    names = ("l1_ptr", "l1_idx", "l1_sub", "l1_top", "l1_cnt", "l2_ptr", "l2_idx", "l2_top", "l2_cnt", "l3_ptr", "l3_idx", "n_valid")
    t = {k: i32({"l1_ptr": n + 1, "l2_ptr": n + 1, "l3_ptr": B + 1, "n_valid": B}.get(k, n)) for k in names}
    nb = lib.nrms_hier_tree_scratch_bytes(B, H)
    scratch = torch.empty(nb // 4 + 2, dtype=torch.float32, device="cuda")
    rc = lib.nrms_hier_tree_build(B, H, _lib.ptr(valid), _lib.ptr(topic), _lib.ptr(sub), *[_lib.ptr(t[k]) for k in names], _lib.ptr(scratch),
                                  C.c_size_t(scratch.numel() * 4), _stream())
    _lib.check(rc, "tree")
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in t.items()}
    seg = lambda ptr, idx, s: list(idx[ptr[s]:ptr[s + 1]])
    assert g["l1_ptr"][0] == 0 and g["l2_ptr"][0] == 0 and g["l3_ptr"][0] == 0
    assert np.all(np.diff(g["l1_ptr"]) >= 0) and np.all(np.diff(g["l2_ptr"]) >= 0) and np.all(np.diff(g["l3_ptr"]) >= 0)
    for b in range(B):
        subs, tops = python_tree(batch["browsed_mask"][b], batch["browsed_categ_ids"][b], batch["browsed_subcateg_ids"][b])
        assert g["n_valid"][b] == int(batch["browsed_mask"][b].sum())
        for s in range(H):
            slot = b * H + s
            if s < len(subs):
                assert sorted(seg(g["l1_ptr"], g["l1_idx"], slot)) == [b * H + k for k in subs[s][2]], (b, s)
                assert (g["l1_sub"][slot], g["l1_top"][slot], g["l1_cnt"][slot]) == (subs[s][0], subs[s][1], len(subs[s][2]))
            else:
                assert seg(g["l1_ptr"], g["l1_idx"], slot) == [] and g["l1_cnt"][slot] == 0
            if s < len(tops):
                assert sorted(seg(g["l2_ptr"], g["l2_idx"], slot)) == [b * H + gi for gi in tops[s][1]], (b, s)
                assert g["l2_top"][slot] == tops[s][0] and g["l2_cnt"][slot] == sum(len(subs[gi][2]) for gi in tops[s][1])
            else:
                assert seg(g["l2_ptr"], g["l2_idx"], slot) == [] and g["l2_cnt"][slot] == 0
        assert sorted(seg(g["l3_ptr"], g["l3_idx"], b)) == [b * H + s for s in range(len(tops))]
    # the matching: slots and click shares of every candidate
    Cn = shape.n_candidates
    ct = torch.from_numpy(batch["candidate_categ_ids"]).cuda()
    cs = torch.from_numpy(batch["candidate_subcateg_ids"]).cuda()
    ss, ts = i32(B * Cn), i32(B * Cn)
    sf, tf = torch.empty(B * Cn, device="cuda"), torch.empty(B * Cn, device="cuda")
    rc = lib.nrms_hier_match(B, Cn, H, _lib.ptr(ct), _lib.ptr(cs), _lib.ptr(t["l1_sub"]), _lib.ptr(t["l1_cnt"]), _lib.ptr(t["l2_top"]),
                             _lib.ptr(t["l2_cnt"]), _lib.ptr(t["n_valid"]), _lib.ptr(ss), _lib.ptr(sf), _lib.ptr(ts), _lib.ptr(tf), _stream())
    _lib.check(rc, "match")
    ss, ts, sf, tf = (x.cpu().numpy().reshape(B, Cn) for x in (ss, ts, sf, tf))
    for b in range(B):
        subs, tops = python_tree(batch["browsed_mask"][b], batch["browsed_categ_ids"][b], batch["browsed_subcateg_ids"][b])
        nv = max(int(batch["browsed_mask"][b].sum()), 1)
        for c in range(Cn):
            want = [i for i, g_ in enumerate(subs) if g_[0] == batch["candidate_subcateg_ids"][b, c]]
            assert ss[b, c] == (b * H + want[0] if want else -1)
            assert sf[b, c] == pytest.approx(len(subs[want[0]][2]) / nv if want else 0.0, abs=1e-7)
            want = [i for i, t_ in enumerate(tops) if t_[0] == batch["candidate_categ_ids"][b, c]]
            assert ts[b, c] == (b * H + want[0] if want else -1)
            assert tf[b, c] == pytest.approx(sum(len(subs[gi][2]) for gi in tops[want[0]][1]) / nv if want else 0.0, abs=1e-7)


CASES = {
    # B, H, C, L, d, heads, q, kwargs of make_batch_hierec
    "small": (6, 12, 4, 8, 64, 4, 32, dict()),
    "empty_user_masked_cands": (5, 20, 5, 6, 64, 4, 32, dict(empty_history_user=True, mask_some_candidates=True)),
    "mixed_topics_h64": (3, 64, 3, 5, 40, 2, 16, dict(consistent_topics=False)),
    "mind_dims": (4, 50, 5, 30, 300, 10, 200, dict()),
}


def _oracle(params, batch, heads, dscores):
    from oracle import nrms_oracle as orc
    from oracle import segpool_oracle as so
    pt = orc.to_torch(params, requires_grad=True)
    s = so.hierec_forward(pt, batch, heads)
    live = torch.as_tensor(batch["candidate_mask"]) != 0
    (torch.where(live, s, torch.zeros_like(s)) * torch.from_numpy(dscores)).sum().backward()
    return s.detach().numpy(), {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape, np.float32)) for k, v in pt.items()}


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_scores_and_every_gradient_against_the_oracle(case, precision):
    B, H, Cn, L, d, h, q, kw = CASES[case]
    shape = synth.Shape(n_words=200, word_embed_size=d, num_attention_heads=h, query_vector_dim=q, batch_size=B, history_len=H,
                        n_candidates=Cn, n_words_title=L)
    params = synth.make_params_hierec(shape, N_SUB, N_TOP, seed=3)
    batch = synth.make_batch_hierec(shape, N_SUB, N_TOP, seed=4, **kw)
    dscores = (np.random.default_rng(5).standard_normal((B, Cn)) * 0.1).astype(np.float32)
    o_scores, o_grads = _oracle(params, batch, h, dscores)
    model = make_hierec(shape, params, precision=precision)
    model.train()
    model.zero_grad()
    scores = model(tbatch(batch))
    scores.backward(torch.from_numpy(dscores).cuda())
    got = scores.detach().cpu().numpy()
    live = batch["candidate_mask"] != 0
    assert np.all(got[~live] == np.float32(-1e9))
    err = float(np.abs(got - o_scores)[live].max())
    scale = float(np.abs(o_scores[live]).max())
    fp16 = precision == "fp16"
    print("hierec %-24s %-6s scores err %.2e (scale %.2f)" % (case, precision, err, scale))
    assert err <= (2e-4 if fp16 else 2e-5) * max(1.0, scale)
    named = dict(model.named_parameters())
    gscale = max(float(np.abs(g).max()) for k, g in o_grads.items() if not k.endswith("word_embedding.0.weight"))
    for n, ref in o_grads.items():
        g = named[n].grad.detach().cpu().numpy()
        if n.endswith("_embedding.weight"):
            ref = ref.copy()                      # F.embedding-free oracle: row 0 (padding_idx) takes no gradient in the model
            g = g.copy()
            ref[0] = 0
            g[0] = 0
        rel = 2e-2 if fp16 else 1e-3
        # (d(W_K.bias) is identically zero in exact arithmetic -- softmax is shift-invariant -- so its bound is the noise term)
        bound = rel * np.abs(ref) + (rel * 0.5) * float(np.abs(ref).max()) + (1e-4 if fp16 else 2e-6) * gscale + 1e-9
        bad = float((np.abs(g - ref) - bound).max())
        print("      %-58s err %.2e  scale %.2e" % (n, float(np.abs(g - ref).max()), float(np.abs(ref).max())))
        assert bad <= 0.0, (case, precision, n, float(np.abs(g - ref).max()), float(np.abs(ref).max()))
    # an eval-mode forward (no saved activations) gives the same scores
    model.eval()
    with torch.no_grad():
        inf = model(tbatch(batch)).cpu().numpy()
    assert float(np.abs(inf - got)[live].max()) <= (2e-4 if fp16 else 1e-6) * max(1.0, scale)


def test_train_step_matches_an_oracle_adam_step_and_is_bit_reproducible():
    from oracle import nrms_oracle as orc
    from oracle import segpool_oracle as so
    B, H, Cn, L, d, h, q, kw = CASES["small"]
    shape = synth.Shape(n_words=200, word_embed_size=d, num_attention_heads=h, query_vector_dim=q, batch_size=B, history_len=H,
                        n_candidates=Cn, n_words_title=L)
    params = synth.make_params_hierec(shape, N_SUB, N_TOP, seed=3)
    batch = synth.make_batch_hierec(shape, N_SUB, N_TOP, seed=4)
    pt = orc.to_torch(params, requires_grad=True)
    loss = orc.loss_fn(so.hierec_forward(pt, batch, h))
    loss.backward()
    want = {}
    for k, v in pt.items():
        g = v.grad.numpy().copy()
        if k.endswith("embedding.weight") or k.endswith("embedding.0.weight"):
            g[0] = 0                               # padding_idx = 0 of all three tables
        want[k] = v.detach().numpy().copy()
        orc.adam_step(want[k], g, np.zeros_like(g), np.zeros_like(g), 1, lr=1e-3)
    outs = []
    for _ in range(2):
        model = make_hierec(shape, params, precision="fp32")
        model.train()
        ls = model.train_step(tbatch(batch), lr=1e-3)
        torch.cuda.synchronize()
        outs.append((float(ls.item()) / B, {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}))
    assert outs[0][0] == pytest.approx(float(loss.detach()), rel=1e-5)
    assert outs[0][0] == outs[1][0]
    for k in want:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k
        # Adam's first step moves every touched element by lr * sign(g): compare where the gradient is clearly non-zero
        moved = np.abs(want[k] - params[k]) > 0.5e-3
        assert float(np.abs(outs[0][1][k] - want[k])[moved].max(initial=0.0)) < 2e-5, k


@pytest.mark.parametrize("model_name", ["hierec", "graph"])
def test_f4_models_against_the_oracle_at_the_benchmarked_size(model_name):
    """bench.py's `variants.hierec` / `variants.graph` configurations (512 users, H = 50, C = 5, 30-word titles, d = 300; fp16 news
    encoder + bf16x3 aggregates) against oracle/segpool_oracle.py on the whole batch: the training forward's kernels with dropout 0.
    PARITY UNPINNED (module docstring): what is pinned is that the timed configuration computes the specification at full size."""
    import time
    from oracle import nrms_oracle as orc
    from oracle import segpool_oracle as so
    shape = synth.Shape(n_words=synth.BENCH.n_words, word_embed_size=300, num_attention_heads=10, query_vector_dim=200, batch_size=512,
                        history_len=50, n_candidates=5, n_words_title=30)
    if model_name == "hierec":
        n_sub, n_top = 286, 19
        params = synth.make_params_hierec(shape, n_sub, n_top, seed=0)
        batch = synth.make_batch_hierec(shape, n_sub, n_top, seed=1, mask_some_candidates=True)
        model = make_hierec(shape, params, precision="fp16", n_sub=n_sub, n_top=n_top)
        oracle = lambda p: so.hierec_forward(p, batch, 10)
    else:
        from tests.test_hip_graph import make_graph
        params = synth.make_params_graph(shape, seed=0)
        batch = synth.make_batch_graph(shape, 8, seed=1, mask_some_candidates=True)
        model = make_graph(shape, params, precision="fp16")
        oracle = lambda p: so.graph_forward(p, batch, 10)
    t0 = time.time()
    with torch.no_grad():
        ref = oracle(orc.to_torch(params)).numpy()
    t_or = time.time() - t0
    model.train()
    live = batch["candidate_mask"] != 0
    scale = float(np.abs(ref[live]).max())
    got = model(tbatch(batch)).detach().cpu().numpy()          # the training forward: fp16 news encoder (config.dropout = 0)
    with torch.no_grad():
        inf = model(tbatch(batch)).cpu().numpy()               # a pass without a backward: precision "fp16" routes it to bf16x3
    for name, s_, bar in (("training forward (fp16 news encoder)", got, 2e-4), ("inference (bf16x3)", inf, 1e-5)):
        err = float(np.abs(s_ - ref)[live].max())
        print("%s at 512 users, %s: max |score - oracle| = %.2e over %d scores (max |score| %.2f; oracle %.0f s)"
              % (model_name, name, err, int(live.sum()), scale, t_or))
        assert np.all(s_[~live] == np.float32(-1e9))
        assert err <= bar * max(1.0, scale), (name, err)
