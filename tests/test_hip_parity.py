"""Parity of the HIP path (through the drop-in Model and the C ABI) against the golden fixtures
generated from the imported reference, and against the oracle on seeded inputs -- in EVERY precision mode
the library offers for training (fp32 and the benchmarked split-bf16 mode; the reduced modes get their own
test further down).

Tolerances, stated once (TOL):
  scores / vectors / loss : absolute (north_star asks 1e-4);
  gradients               : |got - ref| <= g_rtol |ref| + g_atol + g_scale max|ref| per tensor
                            (fp32: summation order only; bf16x3: ~2^-16 relative per product on top).
"""
import os

import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import synth

pytestmark = pytest.mark.gpu

MODES = ["fp32", "bf16x3"]
TOL = {
    "fp32":   dict(score=1e-5, g_rtol=1e-3, g_atol=2e-6, g_scale=0.0),
    "bf16x3": dict(score=2e-5, g_rtol=1e-3, g_atol=2e-6, g_scale=2e-5),     # measured: scores 4e-7 (g2), vectors of norm ~10: 1.1e-5
}
SCORE_TOL = TOL["fp32"]["score"]
GRAD_RTOL, GRAD_ATOL = TOL["fp32"]["g_rtol"], TOL["fp32"]["g_atol"]


def assert_grad_close(got, ref, mode, name):
    t = TOL[mode]
    ref = np.asarray(ref)
    bound = t["g_rtol"] * np.abs(ref) + t["g_atol"] + t["g_scale"] * float(np.abs(ref).max() if ref.size else 0.0)
    diff = np.abs(np.asarray(got) - ref)
    worst = float((diff - bound).max()) if diff.size else 0.0
    assert worst <= 0.0, "%s [%s]: max |diff| %.3e exceeds the bound by %.3e (scale %.3e)" % (
        name, mode, float(diff.max()), worst, float(np.abs(ref).max()))


def make_model(shape, params, dropout=0.0, device="cuda", precision="fp32", fp16_user=False, fp16_inference=True):
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.model.nrms_hip import Model
    cfg = Config("nrms_hip")
    cfg.__nrms__()
    cfg.word_embed_size = shape.word_embed_size
    cfg.num_attention_heads = shape.num_attention_heads
    cfg.query_vector_dim = shape.query_vector_dim
    cfg.dropout = dropout
    cfg.precision = precision
    cfg.fp16_user_encoder = fp16_user          # precision "fp16" only: the user encoder in fp16 too (default: bf16x3)
    # precision "fp16" only: inference passes on the fused fp16 kernels as well, so that the eval-mode tests exercise them
    # (the product default is False: evaluation runs in bf16x3, tests/test_hip_eval.py::test_fp16_mode_evaluates_in_bf16x3)
    cfg.fp16_inference = fp16_inference
    m = Model(cfg, pretrained_word_embedding=params["news_encoder.word_embedding.0.weight"])
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return m.to(device)


def tbatch(batch):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}


def fwd_bwd(model, batch):
    model.zero_grad()
    scores = model(tbatch(batch))
    loss = torch.nn.CrossEntropyLoss()(scores, torch.zeros(len(scores), dtype=torch.long, device=scores.device))
    loss.backward()
    grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}
    return scores.detach().cpu().numpy(), float(loss.detach()), grads


ILL_CONDITIONED = ("W_K.bias", "additive_attention.linear.bias", "user_encoder.additive_attention.attention_query_vector")


def assert_params_close(got, want, name, mode="fp32"):
    """Parameters after a few Adam steps (lr = 1e-3) against the reference stepped by torch.optim.Adam.
    The optimizer kernel itself is exact to 1e-7 on identical gradients (test_adam_step_kernel_alone), so what
    is left is Adam's conditioning: update = lr * m / (sqrt(v) + eps) turns a RELATIVE gradient error r into
    ~r * lr, except where |g| ~ eps = 1e-8, where it amplifies absolute noise by lr / eps.  Two tensors are
    such by construction (documented in DESIGN.md section 1): W_K.bias (analytically zero gradient),
    additive_attention.linear.bias (a cancelling sum, ~1e-2 relative fp32 noise already between the torch-CPU
    oracle and the reference) and, on the 7-slot histories of the g5 shape, the user encoder's query vector
    (d q = sum_s ds_s tanh(.)_s with sum_s ds_s = 0 over near-equal rows: the same cancellation; measured 1.4e-5
    to 5.4e-5 on 2 to 16 of its 32 elements, every other tensor <= 1e-5 bar one or two elements); everything else must agree to the oracle's own bar, 1e-5 (tests/test_oracle_golden.py),
    with a handful of near-zero-gradient elements (at most 3, or 0.2 % of a tensor) allowed up to 30 % of one step."""
    diff = np.abs(got - want)
    if name.endswith(ILL_CONDITIONED):
        assert np.median(diff) < 5e-5, (name, float(np.median(diff)))
        assert diff.max() < 3.1e-3, (name, float(diff.max()))
        return
    n_bad = int((diff > 1e-5).sum())
    print("  adam parity %-60s [%s] max %.2e  >1e-5: %d / %d" % (name, mode, float(diff.max()), n_bad, diff.size))
    assert n_bad <= max(3, int(2e-3 * diff.size)), (name, mode, "elements off by more than 1e-5: %d of %d" % (n_bad, diff.size))
    assert diff.max() < 3e-4, (name, mode, float(diff.max()))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_library_loaded_and_native():
    from pytorch_news_recommender_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.nrms_version()
    assert os.path.basename(_lib.LIB_PATH) == "libnrms_hip.so"


@pytest.mark.parametrize("mode", MODES)
def test_g1_odd_golden(golden_dir, mode):
    g = load(golden_dir, "g1_odd.npz")
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=11, pad_row_zero=False)
    batch = synth.make_batch(shape, seed=12, ragged=True, min_title=1, empty_history_user=True,
                             all_pad_title=True, mask_some_candidates=True)
    model = make_model(shape, params, precision=mode)
    model.train()                      # dropout=0: train mode == eval mode numerically
    scores, loss, grads = fwd_bwd(model, batch)
    SCORE_TOL = TOL[mode]["score"]
    np.testing.assert_allclose(scores, g["scores"], rtol=0, atol=SCORE_TOL)
    assert abs(loss - float(g["loss"])) < SCORE_TOL
    assert (scores[batch["candidate_mask"] == 0] == np.float32(-1e9)).all()
    for n in synth.param_names():
        assert_grad_close(grads[n], g["grad/" + n], mode, n)
    assert not grads["news_encoder.word_embedding.0.weight"][0].any()     # padding_idx = 0
    # helper API (nrms_v0.py:278-312)
    model.eval()
    B, H, L = batch["browsed_titles"].shape
    with torch.no_grad():
        hist = model.get_news_vector(torch.from_numpy(batch["browsed_titles"]).reshape(B * H, L)).view(B, H, -1)
        user = model.get_user_vector(hist)
        pred = model.get_prediction(torch.from_numpy(g["cand"][0]), torch.from_numpy(g["user"][0]))
    np.testing.assert_allclose(hist.cpu().numpy(), g["hist"], atol=SCORE_TOL)
    np.testing.assert_allclose(user.cpu().numpy(), g["user"], atol=SCORE_TOL)
    np.testing.assert_allclose(pred.cpu().numpy(), g["cand"][0] @ g["user"][0], atol=SCORE_TOL)


@pytest.mark.parametrize("mode", MODES)
def test_g2_mind_golden(golden_dir, mode):
    g = load(golden_dir, "g2_mind.npz")
    shape = synth.G2_MIND
    params = synth.make_params(shape, seed=21)
    batch = synth.make_batch(shape, seed=22, ragged=True)
    model = make_model(shape, params, precision=mode)
    scores, loss, grads = fwd_bwd(model, batch)
    SCORE_TOL = TOL[mode]["score"]
    print("g2 [%s]: max |score - reference| = %.3e" % (mode, float(np.abs(scores - g["scores"]).max())))
    np.testing.assert_allclose(scores, g["scores"], rtol=0, atol=SCORE_TOL)
    assert abs(loss - float(g["loss"])) < SCORE_TOL
    emb = "news_encoder.word_embedding.0.weight"
    for n in synth.param_names():
        if n == emb:                    # 95 % of the gradient bytes: sampled rows element-wise + every row's sum
            assert_grad_close(grads[n][g["rows"]], g["grad_rows/" + n], mode, n + "[rows]")
            np.testing.assert_allclose(grads[n].sum(axis=1), g["grad_rowsum/" + n], rtol=1e-3, atol=2e-5)
        else:
            assert_grad_close(grads[n], g["grad/" + n], mode, n)
    # eval forward (no autograd) gives the same scores
    model.eval()
    with torch.no_grad():
        s2 = model(tbatch(batch)).cpu().numpy()
    np.testing.assert_allclose(s2, g["scores"], rtol=0, atol=SCORE_TOL)


@pytest.mark.parametrize("mode", MODES)
def test_train_mode_dropout_replayed_through_oracle(mode):
    """Dropout on (the benchmarked configuration: dropout 0.2, padding tokens skipped): export the kernels' keep
    masks, replay them in the oracle, demand parity of scores and every gradient, the embedding table included.
    Proves forward and backward use the same mask at both sites."""
    from oracle import nrms_oracle as orc
    shape = synth.Shape(n_words=500, word_embed_size=60, num_attention_heads=6, query_vector_dim=32,
                        batch_size=6, history_len=9, n_candidates=4, n_words_title=12)
    params = synth.make_params(shape, seed=5)
    batch = synth.make_batch(shape, seed=6, ragged=True, min_title=2)
    model = make_model(shape, params, dropout=0.2, precision=mode)
    SCORE_TOL = TOL[mode]["score"]
    model.train()
    scores, loss, grads = fwd_bwd(model, batch)
    sv = model.engine._saved
    n_titles = shape.batch_size * (shape.history_len + shape.n_candidates)
    L, d = shape.n_words_title, shape.word_embed_size
    keep = {}
    for site, name in ((0, "embed"), (1, "ctx")):
        k = model.engine.dropout_keep_mask(sv["seed"], site, n_titles * L, 0.2)
        keep[name] = k.cpu().view(n_titles, L, d)
    frac = float(keep["embed"].float().mean())
    assert 0.77 < frac < 0.83, frac
    assert not torch.equal(keep["embed"], keep["ctx"])
    assert model.engine.pad_row_zero is True               # the compact (padding-skipping) path is the one under test
    o_scores, o_loss, o_grads, _ = orc.loss_and_grads(params, batch, shape.num_attention_heads, p_drop=0.2, keep=keep)
    np.testing.assert_allclose(scores, o_scores, rtol=0, atol=SCORE_TOL)
    assert abs(loss - o_loss) < SCORE_TOL
    for n in synth.param_names():
        assert_grad_close(grads[n], o_grads[n], mode, n)
    # a second step draws a different mask
    s2, _, _ = fwd_bwd(model, batch)
    assert np.abs(s2 - scores).max() > 1e-6


@pytest.mark.parametrize("mode", MODES)
def test_g5_fused_train_steps(golden_dir, mode):
    """Model.train_step (HIP fwd + CE + bwd + fused Adam) x3 against the reference model stepped
    by torch.optim.Adam (fixture g5)."""
    g = load(golden_dir, "g5_adam.npz")
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=51)
    model = make_model(shape, params, precision=mode)
    model.train()
    model.config.learning_rate = 1e-3
    losses = []
    for t in range(3):
        batch = synth.make_batch(shape, seed=52 + t, ragged=True, min_title=1)
        ls = model.train_step(tbatch(batch))
        losses.append(float(ls) / shape.batch_size)
    np.testing.assert_allclose(losses, g["losses"], atol=SCORE_TOL)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for n in synth.param_names():
        if n.endswith("W_K.bias"):       # analytically zero gradient: Adam amplifies rounding noise
            assert np.abs(sd[n] - params[n]).max() <= 3.1e-3
            continue
        assert_params_close(sd[n], g["param/" + n], n, mode)


@pytest.mark.parametrize("mode", MODES)
def test_autograd_path_with_torch_adam_matches_fused(golden_dir, mode):
    """The drop-in loop of train_eval.py:111-127 (model(batch) -> CE -> backward -> torch Adam)."""
    g = load(golden_dir, "g5_adam.npz")
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=51)
    model = make_model(shape, params, precision=mode)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    losses = []
    for t in range(3):
        batch = synth.make_batch(shape, seed=52 + t, ragged=True, min_title=1)
        out = model(tbatch(batch))
        model.zero_grad()
        loss = crit(out, torch.zeros(len(out), dtype=torch.long, device=out.device))
        losses.append(loss.item())
        loss.backward()
        opt.step()
    np.testing.assert_allclose(losses, g["losses"], atol=SCORE_TOL)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for n in synth.param_names():
        if n.endswith("W_K.bias"):
            continue
        assert_params_close(sd[n], g["param/" + n], n, mode)


def test_adam_step_kernel_alone():
    """nrms_adam_step on its own, identical gradients on both sides, against oracle.adam_step (numpy fp32
    restatement of torch.optim.Adam, pinned by fixture g5): 4 steps, also with grad_scale (the 1/world_size of
    the data-parallel path).  Bar: 1e-7 relative to the parameter scale -- what is left of the g5 tolerance
    after this is the gradients' conditioning, not the optimizer."""
    from oracle import nrms_oracle as orc
    model = make_model(synth.G1_ODD, synth.make_params(synth.G1_ODD, seed=1))
    eng = model.engine
    rng = np.random.default_rng(3)
    n = 100_003                                     # not a multiple of 4: exercises the tail of the float4 pass
    p0 = rng.normal(0, 0.3, n).astype(np.float32)
    grads = [(rng.normal(0, 1, n) * 10.0 ** rng.uniform(-9, -1, n)).astype(np.float32) for _ in range(4)]
    for gs in (1.0, 0.25):
        p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        dp, dm, dv = (torch.from_numpy(a.copy()).cuda() for a in (p0, m, v))
        for t, g in enumerate(grads, start=1):
            orc.adam_step(p, (g * np.float32(gs)).astype(np.float32), m, v, t, lr=1e-3)
            eng.adam_step(dp, torch.from_numpy(g).cuda(), dm, dv, t, lr=1e-3, grad_scale=gs)
        err = np.abs(dp.cpu().numpy() - p)
        print("adam alone (grad_scale %g): max |dparam| = %.3e" % (gs, float(err.max())))
        # torch's own formulation: the update is lr * m_hat / (sqrt(v_hat) + eps) <= ~lr per step; agreement to
        # 1e-7 absolute on parameters of scale 0.3 is ~1 ulp
        assert err.max() <= 1.5e-7, float(err.max())
        np.testing.assert_allclose(dm.cpu().numpy(), m, rtol=2e-6, atol=2e-7 * float(np.abs(m).max()))
        np.testing.assert_allclose(dv.cpu().numpy(), v, rtol=4e-6, atol=2e-7 * float(v.max()))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape", [
    synth.Shape(n_words=300, word_embed_size=64, num_attention_heads=2, query_vector_dim=64,
                batch_size=5, history_len=33, n_candidates=2, n_words_title=33),     # S > 32, d_k = 32
    synth.Shape(n_words=300, word_embed_size=300, num_attention_heads=6, query_vector_dim=200,
                batch_size=3, history_len=64, n_candidates=3, n_words_title=20),     # v1 heads: d_k = 50, H = 64
    synth.Shape(n_words=64, word_embed_size=8, num_attention_heads=2, query_vector_dim=4,
                batch_size=1, history_len=1, n_candidates=1, n_words_title=1),       # minimum sizes
    synth.Shape(n_words=1000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                batch_size=16, history_len=50, n_candidates=5, n_words_title=30),    # bench shape, small batch
    synth.Shape(n_words=300, word_embed_size=100, num_attention_heads=2, query_vector_dim=32,
                batch_size=3, history_len=6, n_candidates=2, n_words_title=40),      # 40 words, d_k = 50: the two-wave 64 x 64
                                                                                     # attention on the padding-skipping path
])
def test_shape_sweep_against_oracle(shape, mode):
    from oracle import nrms_oracle as orc
    params = synth.make_params(shape, seed=101)
    batch = synth.make_batch(shape, seed=102, ragged=True, min_title=1, mask_some_candidates=shape.n_candidates > 1)
    model = make_model(shape, params, precision=mode)
    scores, loss, grads = fwd_bwd(model, batch)
    o_scores, o_loss, o_grads, _ = orc.loss_and_grads(params, batch, shape.num_attention_heads)
    SCORE_TOL = TOL[mode]["score"]
    np.testing.assert_allclose(scores, o_scores, rtol=0, atol=SCORE_TOL)
    assert abs(loss - o_loss) < SCORE_TOL
    for n in synth.param_names():
        assert_grad_close(grads[n], o_grads[n], mode, n)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["all_padding", "no_padding", "one_live_token", "nonzero_pad_row"])
def test_live_row_compaction_edges(case, mode):
    """The X-gradient GEMM and the embedding scatter run on the compacted non-padding token rows
    (padding_idx rows get no gradient, nrms_v0.py:134-136): empty list, full list, a single live row,
    and a table whose row 0 is NOT zero (from_pretrained keeps it: pad tokens then carry signal forward
    but still receive no gradient)."""
    from oracle import nrms_oracle as orc
    shape = synth.Shape(n_words=300, word_embed_size=60, num_attention_heads=6, query_vector_dim=32,
                        batch_size=5, history_len=9, n_candidates=4, n_words_title=11)
    params = synth.make_params(shape, seed=111, pad_row_zero=(case != "nonzero_pad_row"))
    batch = synth.make_batch(shape, seed=112, ragged=(case != "no_padding"), min_title=1)
    if case == "all_padding":
        batch["browsed_titles"][:] = 0
        batch["candidate_titles"][:] = 0
    if case == "one_live_token":
        batch["browsed_titles"][:] = 0
        batch["candidate_titles"][:] = 0
        batch["candidate_titles"][3, 2, 0] = 17
    if case == "no_padding":
        assert (batch["candidate_titles"] != 0).all() and (batch["browsed_titles"] != 0).all()
    model = make_model(shape, params, precision=mode)
    scores, loss, grads = fwd_bwd(model, batch)
    o_scores, o_loss, o_grads, _ = orc.loss_and_grads(params, batch, shape.num_attention_heads)
    np.testing.assert_allclose(scores, o_scores, rtol=0, atol=TOL[mode]["score"])
    for n in synth.param_names():
        assert_grad_close(grads[n], o_grads[n], mode, n)
    table_grad = grads["news_encoder.word_embedding.0.weight"]
    assert (table_grad[0] == 0).all()                       # padding_idx row
    if case == "all_padding":
        assert (table_grad == 0).all()
    if case == "one_live_token":
        assert (np.abs(table_grad).sum(axis=1) != 0).sum() <= 1


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_padding_token_skip_equals_dense_path(precision):
    """NRMS_FLAG_PAD_ROW_ZERO (embedding row 0 all zeros): the compact path (Q|K|V projection and d(w_qkv) on
    the non-padding tokens only) must reproduce the dense path -- Q|K|V bit for bit (row-independent
    arithmetic; a padding row is 0.w + b = b exactly), scores and gradients up to summation order -- also in train mode
    with both dropouts on (same seed => same masks)."""
    shape = synth.Shape(n_words=500, word_embed_size=120, num_attention_heads=6, query_vector_dim=64,
                        batch_size=12, history_len=20, n_candidates=4, n_words_title=17)
    params = synth.make_params(shape, seed=131)            # row 0 of the table is zero
    batch = synth.make_batch(shape, seed=132, ragged=True, min_title=1, all_pad_title=True, empty_history_user=True)
    out = {}
    for skip in (True, False):
        model = make_model(shape, params, dropout=0.2)
        model.config.precision = precision
        model.config.skip_padding_tokens = skip
        model.train()
        torch.manual_seed(7)
        model._calls = 0
        scores, loss, grads = fwd_bwd(model, batch)
        assert model.engine.pad_row_zero is skip
        out[skip] = (scores, grads)
    # (an all-padding title takes the closed form ctx = b_v instead of summing 17 equal terms b_v / 17)
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=0, atol=1e-6)
    # gradients: to summation order, measured against the overall gradient scale (several tensors -- W_K.bias,
    # the user encoder's additive weights at this size, d(b_add) as a cancelling sum -- are pure rounding
    # noise many orders below it, where a relative comparison says nothing)
    gscale = max(float(np.abs(out[False][1][n]).max()) for n in synth.param_names()
                 if not n.endswith("word_embedding.0.weight"))
    for n in synth.param_names():
        a, b = out[True][1][n], out[False][1][n]
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-7 * gscale, err_msg=n)
    # a table with a non-zero padding row never takes the compact path
    params2 = synth.make_params(shape, seed=131, pad_row_zero=False)
    model = make_model(shape, params2)
    fwd_bwd(model, batch)
    assert model.engine.pad_row_zero is False


@pytest.mark.parametrize("precision,score_tol,grad_rtol", [("bf16", 5e-3, 6e-2)])
def test_reduced_precision_modes(golden_dir, precision, score_tol, grad_rtol):
    """Plain bf16 projections against the REFERENCE fixture: reported with a loose bound (it cannot meet
    north_star's 1e-4 score bar and is never a default).  bf16x3 is covered like fp32 by every test above."""
    g = load(golden_dir, "g2_mind.npz")
    shape = synth.G2_MIND
    params = synth.make_params(shape, seed=21)
    batch = synth.make_batch(shape, seed=22, ragged=True)
    model = make_model(shape, params)
    model.config.precision = precision
    scores, loss, grads = fwd_bwd(model, batch)
    err = float(np.abs(scores - g["scores"]).max())
    print("precision %s: max |score - reference| = %.3e, |loss diff| = %.3e" % (precision, err, abs(loss - float(g["loss"]))))
    assert err < score_tol
    assert abs(loss - float(g["loss"])) < score_tol
    for n in synth.param_names():
        if n.endswith("word_embedding.0.weight") or n.endswith("W_K.bias"):
            continue
        ref = g["grad/" + n]
        scale = np.abs(ref).max()
        assert np.abs(grads[n] - ref).max() <= grad_rtol * scale + 2e-6, n


def test_out_of_range_word_id_is_reported():
    """nn.Embedding raises on an index outside the table (nrms_v0.py:134-139); the HIP path replaces such ids by
    the padding id on the device (no kernel indexes out of bounds) and raises NrmsError at the next check."""
    from pytorch_news_recommender_amd._lib import NrmsError
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=11)
    model = make_model(shape, params).eval()
    good = synth.make_batch(shape, seed=12, ragged=True)
    with torch.no_grad():
        ref = model(tbatch(good)).cpu().numpy()
    model.engine.check_ids()                                      # clean so far
    for bad_id in (shape.n_words, -1, 2 ** 40):
        batch = {k: v.copy() for k, v in good.items()}
        batch["candidate_titles"][1, 0, 0] = bad_id
        batch["browsed_titles"][0, 1, 2] = bad_id
        with torch.no_grad():
            s = model(tbatch(batch))
        torch.cuda.synchronize()
        assert np.isfinite(s.cpu().numpy()[batch["candidate_mask"] == 1]).all()
        with pytest.raises(NrmsError, match="outside"):
            model.engine.check_ids()
        model.engine.check_ids()                                  # the count is consumed by the raise
    # training path + deferred (non-blocking) report on a later call
    model.train()
    batch = {k: v.copy() for k, v in good.items()}
    batch["browsed_titles"][0, 0, 0] = shape.n_words + 5
    model.train_step(tbatch(batch))
    torch.cuda.synchronize()
    with pytest.raises(NrmsError, match="outside"):
        model.train_step(tbatch(good))
    model.engine.check_ids()                                      # reported once; clean input stays clean
    model.eval()
    with torch.no_grad():
        assert np.isfinite(model(tbatch(good)).cpu().numpy()).all()
    model.engine.check_ids()
    assert ref.shape == (shape.batch_size, shape.n_candidates)


@pytest.mark.parametrize("mode", MODES + ["fp16"])
def test_int32_word_ids_give_the_same_results(mode):
    """The C ABI validates int32 or int64 ids into its own int64 copy (nrms_sanitize_ids / _i32): a feed that keeps its ids
    in 32 bits gets bit-identical scores, gradients and error reporting."""
    from pytorch_news_recommender_amd._lib import NrmsError
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=11)
    batch = synth.make_batch(shape, seed=12, ragged=True, mask_some_candidates=True)
    b64 = tbatch(batch)
    b32 = {k: (v.to(torch.int32) if k.endswith("titles") else v) for k, v in b64.items()}
    model = make_model(shape, params, precision=mode).train()
    s64, l64, g64 = fwd_bwd(model, batch)
    model.zero_grad()
    scores = model(b32)
    loss = torch.nn.CrossEntropyLoss()(scores, torch.zeros(len(scores), dtype=torch.long, device=scores.device))
    loss.backward()
    assert np.array_equal(scores.detach().cpu().numpy(), s64)
    t = "news_encoder.word_embedding.0.weight"
    assert np.array_equal(dict(model.named_parameters())[t].grad.cpu().numpy(), g64[t])
    model.eval()
    with torch.no_grad():
        assert torch.equal(model(b32), model(b64))                 # the de-duplicating inference path too
    b32["candidate_titles"] = b32["candidate_titles"].clone()
    b32["candidate_titles"][0, 0, 0] = -7
    with torch.no_grad():
        model(b32)
    with pytest.raises(NrmsError, match="outside"):
        model.engine.check_ids()


def test_second_training_forward_invalidates_the_first_backward():
    """One slot of saved activations: a backward for an earlier training forward must raise, not run on the
    activations of a later one (the reference's autograd would keep both graphs).  Inference calls in between
    (get_news_vector, an eval forward) use their own buffers and leave the pending backward intact."""
    from pytorch_news_recommender_amd._lib import NrmsError
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=11)
    batch = synth.make_batch(shape, seed=12, ragged=True)
    model = make_model(shape, params).train()
    s_ref, _, g_ref = fwd_bwd(model, batch)
    model.zero_grad()
    s1 = model(tbatch(batch))
    with torch.no_grad():                                         # inference between forward and backward
        model.get_news_vector(torch.from_numpy(batch["candidate_titles"][:, 0]))
        model.eval()
        model(tbatch(synth.make_batch(shape, seed=99, ragged=True)))
        model.train()
    torch.nn.functional.cross_entropy(s1, torch.zeros(len(s1), dtype=torch.long, device=s1.device)).backward()
    for n, p in model.named_parameters():
        np.testing.assert_array_equal(p.grad.cpu().numpy(), g_ref[n], err_msg=n)
    model.zero_grad()
    s1 = model(tbatch(batch))
    s2 = model(tbatch(synth.make_batch(shape, seed=98, ragged=True)))
    with pytest.raises(NrmsError, match="replaced"):
        s1.sum().backward()
    s2.sum().backward()                                           # the latest forward still backpropagates


def test_a_retained_grad_tensor_keeps_its_values():
    """torch guarantees that a tensor the caller kept from an earlier backward (a saved p.grad, a hook's argument) is not
    overwritten by a later one.  The autograd path therefore hands out a fresh flat gradient buffer per backward unless
    the caller opted into reuse (model.reuse_grad_buffer, set by train_eval.train for the reference's loop)."""
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=11)
    model = make_model(shape, params).train()
    name = "news_encoder.multihead_self_attention.W_V.weight"
    p = dict(model.named_parameters())[name]
    _, _, g1 = fwd_bwd(model, synth.make_batch(shape, seed=12, ragged=True))
    kept = p.grad                                   # no clone: the very tensor autograd installed
    model.zero_grad(set_to_none=True)
    _, _, g2 = fwd_bwd(model, synth.make_batch(shape, seed=13, ragged=True))
    assert not np.array_equal(g1[name], g2[name])
    np.testing.assert_array_equal(kept.cpu().numpy(), g1[name])
    # opt-in reuse: the same storage serves every backward (what the reference-shaped training loop wants)
    model.reuse_grad_buffer = True
    model.zero_grad(set_to_none=True)
    fwd_bwd(model, synth.make_batch(shape, seed=12, ragged=True))
    ptr = p.grad.data_ptr()
    model.zero_grad(set_to_none=True)
    fwd_bwd(model, synth.make_batch(shape, seed=13, ragged=True))
    assert p.grad.data_ptr() == ptr


def test_two_engines_on_two_streams_in_one_process():
    """Helper streams are per caller stream (capi.hip side_streams_for): two models trained from two host threads on two HIP
    streams of one process, their backwards in flight at the same time, give exactly the gradients each gives alone."""
    import threading
    shape = synth.Shape(n_words=3000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=16, history_len=50, n_candidates=5, n_words_title=30)
    jobs = []
    for i, prec in enumerate(("fp16", "bf16x3")):
        params = synth.make_params(shape, seed=70 + i)
        batch = synth.make_batch(shape, seed=80 + i, ragged=True, min_title=1)
        model = make_model(shape, params, precision=prec).train()
        alone = fwd_bwd(model, batch)
        jobs.append(dict(model=model, batch=batch, alone=alone, stream=torch.cuda.Stream(), out=None, err=None))
    torch.cuda.synchronize()

    def run(job):
        try:
            with torch.cuda.stream(job["stream"]):
                for _ in range(4):                  # several steps each, so that the two threads really overlap
                    job["out"] = fwd_bwd(job["model"], job["batch"])
            job["stream"].synchronize()
        except Exception as e:                      # surfaced below
            job["err"] = e

    threads = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for j in jobs:
        assert j["err"] is None, j["err"]
        np.testing.assert_array_equal(j["out"][0], j["alone"][0])
        for n in j["alone"][2]:
            np.testing.assert_array_equal(j["out"][2][n], j["alone"][2][n], err_msg=n)
