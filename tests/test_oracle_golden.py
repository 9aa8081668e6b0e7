"""The oracle (oracle/nrms_oracle.py) against the fixtures produced by the imported
reference (tests/golden/gen_golden.py).  This is the pin that lets the oracle stand in
for the reference on the GPU box, where /root/reference does not exist."""
import os

import numpy as np
import pytest
import torch

from oracle import nrms_oracle as orc
from pytorch_news_recommender_amd import synth

TOL = 2e-6   # fp32 vs fp32, different summation order only


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_g1_odd_forward_backward(golden_dir):
    g = load(golden_dir, "g1_odd.npz")
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=11, pad_row_zero=False)
    batch = synth.make_batch(shape, seed=12, ragged=True, min_title=1, empty_history_user=True,
                             all_pad_title=True, mask_some_candidates=True)
    # fixture really contains the edge cases
    assert batch["browsed_lens"][1] == 0 and not batch["browsed_titles"][0, 0].any()
    assert (batch["candidate_mask"] == 0).sum() == 3
    for per_slot in (False, True):
        scores, loss, grads, aux = orc.loss_and_grads(params, batch, shape.num_attention_heads,
                                                      per_slot=per_slot)
        np.testing.assert_allclose(scores, g["scores"], rtol=0, atol=TOL)
        assert abs(loss - float(g["loss"])) < TOL
        np.testing.assert_allclose(aux["hist"], g["hist"], atol=TOL)
        np.testing.assert_allclose(aux["cand"], g["cand"], atol=TOL)
        np.testing.assert_allclose(aux["user"], g["user"], atol=TOL)
        for n in synth.param_names():
            np.testing.assert_allclose(grads[n], g["grad/" + n], rtol=1e-4, atol=TOL, err_msg=n)
    # padding_idx=0: row 0 is used in forward (non-zero here) but gets no gradient
    assert np.abs(params["news_encoder.word_embedding.0.weight"][0]).max() > 0
    assert not g["grad/news_encoder.word_embedding.0.weight"][0].any()
    # masked candidates sit at exactly -1e9
    assert (g["scores"][batch["candidate_mask"] == 0] == np.float32(-1e9)).all()


def test_g1_fp64_oracle_close(golden_dir):
    """The fp64 oracle is the high-precision truth used to rank fp32/bf16 HIP errors."""
    g = load(golden_dir, "g1_odd.npz")
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=11, pad_row_zero=False)
    batch = synth.make_batch(shape, seed=12, ragged=True, min_title=1, empty_history_user=True,
                             all_pad_title=True, mask_some_candidates=True)
    scores, loss, _, _ = orc.loss_and_grads(params, batch, shape.num_attention_heads, dtype=torch.float64)
    np.testing.assert_allclose(scores, g["scores"], atol=2e-6)
    assert abs(loss - float(g["loss"])) < 2e-6


def test_g2_mind_shape(golden_dir):
    g = load(golden_dir, "g2_mind.npz")
    shape = synth.G2_MIND
    params = synth.make_params(shape, seed=21)
    batch = synth.make_batch(shape, seed=22, ragged=True)
    scores, loss, grads, aux = orc.loss_and_grads(params, batch, shape.num_attention_heads)
    np.testing.assert_allclose(scores, g["scores"], atol=TOL)
    assert abs(loss - float(g["loss"])) < TOL
    np.testing.assert_allclose(aux["user"], g["user"], atol=TOL)
    np.testing.assert_allclose(aux["hist"], g["hist"], atol=TOL)
    np.testing.assert_allclose(aux["cand"], g["cand"], atol=TOL)
    emb = "news_encoder.word_embedding.0.weight"
    for n in synth.param_names():
        if n == emb:
            np.testing.assert_allclose(grads[n][g["rows"]], g["grad_rows/" + n], rtol=1e-4, atol=TOL)
            np.testing.assert_allclose(grads[n].sum(axis=1), g["grad_rowsum/" + n], rtol=1e-3, atol=2e-5)
        else:
            np.testing.assert_allclose(grads[n], g["grad/" + n], rtol=1e-4, atol=TOL, err_msg=n)


def test_g3_v1_semantics(golden_dir):
    g = load(golden_dir, "g3_v1.npz")
    rng = np.random.default_rng(31)
    N, S, d, h, q = 5, 11, 300, 6, 200
    X = rng.normal(0, 0.5, size=(N, S, d)).astype(np.float32)
    lens = np.array([11, 7, 1, 4, 9])
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.uint8)
    p = {}
    for n in ("W_Q", "W_K", "W_V"):
        p["m." + n + ".weight"] = rng.uniform(-0.1, 0.1, size=(d, d)).astype(np.float32)
        p["m." + n + ".bias"] = rng.uniform(-0.05, 0.05, size=(d,)).astype(np.float32)
    p["m.W_O.weight"] = rng.uniform(-0.1, 0.1, size=(d, d)).astype(np.float32)
    p["m.W_O.bias"] = rng.uniform(-0.05, 0.05, size=(d,)).astype(np.float32)
    p["a.linear.weight"] = rng.uniform(-0.1, 0.1, size=(q, d)).astype(np.float32)
    p["a.linear.bias"] = rng.uniform(-0.05, 0.05, size=(q,)).astype(np.float32)
    p["a.attention_query_vector"] = rng.uniform(-0.1, 0.1, size=(q,)).astype(np.float32)
    pt = orc.to_torch(p)
    Xt, mt = torch.from_numpy(X), torch.from_numpy(mask)
    np.testing.assert_allclose(orc.multihead_self_attention(pt, "m.", Xt, h).numpy(), g["mhsa_nomask"], atol=TOL)
    np.testing.assert_allclose(orc.multihead_self_attention(pt, "m.", Xt, h, mask=mt).numpy(), g["mhsa_mask"], atol=TOL)
    np.testing.assert_allclose(orc.additive_attention(pt, "a.", Xt).numpy(), g["add_nomask"], atol=TOL)
    np.testing.assert_allclose(orc.additive_attention(pt, "a.", Xt, mt).numpy(), g["add_mask"], atol=TOL)


def test_g4_auc(golden_dir):
    g = load(golden_dir, "g4_auc.npz")
    scores, labels = synth.make_eval_impressions(n_imp=40, max_cand=300, seed=7)
    mean, aucs = orc.mean_impression_auc(scores, labels)
    np.testing.assert_allclose(aucs, g["aucs"], rtol=0, atol=1e-12)
    assert abs(mean - float(g["mean"])) < 1e-12
    with pytest.raises(ValueError):
        orc.roc_auc([1, 1, 1], [0.1, 0.2, 0.3])


def test_g5_adam_three_steps(golden_dir):
    g = load(golden_dir, "g5_adam.npz")
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=51)
    batches = [synth.make_batch(shape, seed=52 + t, ragged=True, min_title=1) for t in range(3)]
    out, losses = orc.train_steps(params, batches, shape.num_attention_heads, lr=1e-3)
    np.testing.assert_allclose(losses, g["losses"], atol=TOL)
    for n in synth.param_names():
        if n.endswith("W_K.bias"):
            # dL/d(b_K) is analytically 0 (adding q.b_K to every key of a query leaves the
            # softmax unchanged), so the reference's own gradient there is rounding noise
            # (~1e-10) that Adam normalises to +-lr steps: only the 3*lr bound is meaningful.
            assert np.abs(out[n] - params[n]).max() <= 3.1e-3
            assert np.abs(g["param/" + n] - params[n]).max() <= 3.1e-3
            continue
        # Adam's first steps move a weight by ~lr whatever |g|; compare at 1% of one step
        np.testing.assert_allclose(out[n], g["param/" + n], rtol=0, atol=1e-5, err_msg=n)


def test_dropout_keep_mask_semantics():
    """Explicit keep masks reproduce torch dropout scaling: y = x*keep/(1-p)."""
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=3)
    batch = synth.make_batch(shape, seed=4)
    B, H, C, L, d = shape.batch_size, shape.history_len, shape.n_candidates, shape.n_words_title, shape.word_embed_size
    n = B * (H + C)
    ones = {"embed": torch.ones(n, L, d), "ctx": torch.ones(n, L, d)}
    p = orc.to_torch(params)
    s0, _ = orc.forward(p, batch, shape.num_attention_heads, p_drop=0.0)
    s1, _ = orc.forward(p, batch, shape.num_attention_heads, p_drop=0.5, keep=ones)
    assert not torch.allclose(s0, s1)      # all-keep at p=0.5 doubles activations twice
    zeros = {"embed": torch.zeros(n, L, d), "ctx": torch.ones(n, L, d)}
    s2, aux = orc.forward(p, batch, shape.num_attention_heads, p_drop=0.5, keep=zeros)
    assert torch.isfinite(s2).all()


def naml_sample_rows(n_rows, k=32, seed=77):
    """Same rows as tests/golden/gen_golden.py:naml_sample_rows."""
    return np.sort(np.random.default_rng(seed).choice(n_rows, size=min(k, n_rows), replace=False))


def check_naml_grads(g, tag, grads, names, rtol, atol, scale_floor=0.0):
    """Gradients against fixture g7: small tensors in full, large matrices by sampled rows, row sums and column sums."""
    for n in names:
        got = grads[n]
        if tag + "/grad/" + n in g:
            want = g[tag + "/grad/" + n]
            tol = atol + scale_floor * float(np.abs(want).max())
            np.testing.assert_allclose(got, want, rtol=rtol, atol=tol, err_msg=n)
        else:
            want = g[tag + "/grad_rows/" + n]
            tol = atol + scale_floor * float(np.abs(want).max())
            np.testing.assert_allclose(got[naml_sample_rows(got.shape[0])], want, rtol=rtol, atol=tol, err_msg=n)
            for axis, key in ((1, "grad_rowsum"), (0, "grad_colsum")):
                w = g["%s/%s/%s" % (tag, key, n)]
                np.testing.assert_allclose(got.sum(axis, dtype=np.float64), w, rtol=rtol,
                                           atol=(atol + scale_floor * float(np.abs(w).max())) * 30, err_msg=n + " " + key)


@pytest.mark.parametrize("tag", ["odd", "mind"])
def test_g7_naml_forward_backward(golden_dir, tag):
    """oracle/naml_oracle.py against the imported nrms_naml.Model (SURVEY f-3)."""
    from oracle import naml_oracle as nml
    g = load(golden_dir, "g7_naml.npz")
    shape = synth.G7_ODD if tag == "odd" else synth.G7_MIND
    params = synth.make_params_naml(shape, seed=21)
    batch = synth.make_batch_naml(shape, seed=22)
    assert batch["browsed_lens"][1] == 0 and not batch["browsed_absts"][0, 0].any() and batch["candidate_mask"][0, -1] == 0
    p = nml.to_torch(params)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    scores, parts = nml.forward(p, tb, shape.title_heads_num, shape.user_heads_num, parts=True)
    np.testing.assert_allclose(scores.numpy(), g[tag + "/scores"], rtol=0, atol=2e-5)
    for k in ("cand", "hist", "user"):
        np.testing.assert_allclose(parts[k].numpy(), g[tag + "/" + k], rtol=0, atol=5e-6, err_msg=k)
    scores, loss, grads = nml.loss_and_grads(params, batch, shape.title_heads_num, shape.user_heads_num)
    assert abs(loss - float(g[tag + "/loss"])) < 5e-6
    assert (scores[batch["candidate_mask"] == 0] == np.float32(-1e9)).all()
    check_naml_grads(g, tag, grads, list(params), rtol=1e-4, atol=TOL, scale_floor=2e-6)   # fp32 rounding of a tensor's scale
    for n in ("news_encoder.word_embedding.weight", "news_encoder.category_embedding.weight",
              "news_encoder.subcategory_embedding.weight"):
        assert not grads[n][0].any(), n                       # padding_idx = 0 (nrms_naml.py:107-111)
