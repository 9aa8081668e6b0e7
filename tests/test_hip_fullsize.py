"""Size-independent properties of the HIP path at BASELINE.json's full single-GPU size
(512 users, hist=50, cand=5, title_len=30, d=300, V=45800), where the CPU oracle would take minutes:
determinism, user-permutation equivariance, linearity of the backward in dscores, masked-slot and
padding-row invariants, and agreement of the three precision modes."""
import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from tests.test_hip_parity import make_model
    shape = synth.BENCH
    params = synth.make_params(shape, seed=0)
    batch = synth.make_batch(shape, seed=1, mask_some_candidates=True)
    model = make_model(shape, params)        # dropout 0
    dev = torch.device("cuda")
    tb = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    return shape, model, batch, tb


def _scores(model, tb, training=True):
    eng = model.engine
    return eng.forward(model._flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=training)


def test_forward_is_deterministic_and_masks_hold(full):
    shape, model, batch, tb = full
    s1 = _scores(model, tb).clone()
    s2 = _scores(model, tb)
    assert torch.equal(s1, s2)                                   # no atomics anywhere in the forward
    s = s1.cpu().numpy()
    assert np.isfinite(s[batch["candidate_mask"] == 1]).all()
    assert (s[batch["candidate_mask"] == 0] == np.float32(-1e9)).all()
    assert s.shape == (shape.batch_size, shape.n_candidates)


def test_users_are_independent_permutation_equivariance(full):
    shape, model, batch, tb = full
    perm = torch.from_numpy(np.random.default_rng(3).permutation(shape.batch_size)).cuda()
    s = _scores(model, tb).clone()
    tb2 = {k: v[perm] for k, v in tb.items()}
    s_perm = _scores(model, tb2)
    # a user's scores do not depend on its position in the batch or on the other users: bit-exact
    assert torch.equal(s_perm, s[perm])


def test_backward_is_linear_in_dscores_and_pad_row_gets_no_gradient(full):
    shape, model, batch, tb = full
    eng, flat = model.engine, model._flat
    s = _scores(model, tb)
    g = torch.Generator(device="cpu").manual_seed(5)
    d1 = (torch.randn(s.shape, generator=g) * 1e-3).cuda()
    grads = []
    for scale in (1.0, 2.0):
        _scores(model, tb)
        gf = torch.zeros_like(flat)
        eng.backward(flat, gf, d1 * scale)
        grads.append(gf)
    g1, g2 = grads
    # scaling by 2 is exact in fp32; only the float-atomic scatter of the embedding gradient may reorder sums
    lay = model._layout
    for name in lay.names:
        a, b = lay.view(g1, name), lay.view(g2, name)
        tol = 2e-6 * float(a.abs().max()) + 1e-12
        assert float((2 * a - b).abs().max()) <= tol, name
    emb = lay.view(g1, "news_encoder.word_embedding.0.weight")
    assert not emb[0].any()                                      # padding_idx = 0
    ids = torch.cat([tb["browsed_titles"].reshape(-1), tb["candidate_titles"].reshape(-1)])
    untouched = torch.ones(shape.n_words, dtype=torch.bool, device="cuda")
    untouched[ids.unique()] = False
    assert not emb[untouched].any()                              # rows no title uses stay exactly zero
    # masked candidates pass no gradient: zeroing their dscores changes nothing
    _scores(model, tb)
    gf3 = torch.zeros_like(flat)
    eng.backward(flat, gf3, d1 * tb["candidate_mask"].float())
    for name in lay.names:
        a, b = lay.view(g1, name), lay.view(gf3, name)
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-12, name


def test_precision_modes_agree_at_full_size(full):
    shape, model, batch, tb = full
    ref = _scores(model, tb, training=False).clone()
    valid = tb["candidate_mask"] == 1
    try:
        for prec, lo, tol in (("bf16x3", 1e-9, 1e-4), ("bf16", 1e-5, 5e-3)):
            model.config.precision = prec        # model.engine re-applies config.precision on every access
            assert model.engine.precision == prec
            s = _scores(model, tb, training=False)
            err = float((s - ref)[valid].abs().max())
            print("full size %s vs fp32: max |dscore| = %.3e" % (prec, err))
            assert lo < err < tol                # different arithmetic (not silently the fp32 path), inside the bar
    finally:
        model.config.precision = "fp32"


def test_train_step_runs_and_loss_decreases_on_a_fixed_batch(full):
    shape, model, batch, tb = full
    model.config.learning_rate = 1e-3
    losses = [float(model.train_step(tb)) / shape.batch_size for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_padding_token_skip_equals_dense_path_at_full_size(full):
    """NRMS_FLAG_PAD_ROW_ZERO at the bench shape (844 800 token rows, ~65 % padding, ~41 % all-padding
    titles): scores and every gradient tensor of the compact path against the dense path of the same model,
    and the weight gradients of the compact path are run-to-run reproducible (deterministic compaction)."""
    shape, model, batch, tb = full
    eng, flat = model.engine, model._flat
    lay = model._layout
    g = torch.Generator(device="cpu").manual_seed(9)
    res = {}
    for skip in (True, False, True):
        eng.pad_row_zero = skip
        s = _scores(model, tb).clone()
        d1 = (torch.randn(s.shape, generator=torch.Generator().manual_seed(9)) * 1e-3).cuda()
        gf = torch.zeros_like(flat)
        eng.backward(flat, gf, d1)
        res.setdefault(skip, []).append((s, gf))
    eng.pad_row_zero = model._pad_zero
    (s_a, g_a), (s_c, g_c) = res[True]
    s_b, g_b = res[False][0]
    assert torch.equal(s_a, s_c)
    table = "news_encoder.word_embedding.0.weight"
    for name in lay.names:                       # every tensor, the embedding table included: the compaction is
        assert torch.equal(lay.view(g_a, name), lay.view(g_c, name)), name       # ordered, the scatter atomic-free
    assert float((s_a - s_b).abs().max()) < 2e-6
    gscale = max(float(lay.view(g_b, n).abs().max()) for n in lay.names if n != table)
    for name in lay.names:
        a, b = lay.view(g_a, name), lay.view(g_b, name)
        tol = 1e-4 * float(b.abs().max()) + 2e-7 * gscale
        assert float((a - b).abs().max()) <= tol, (name, float((a - b).abs().max()), tol)


def noise_only(name):
    """Tensors whose fp16-mode gradient at these (freshly initialised) weights is rounding noise, DESIGN section 2: W_K.bias
    is analytically zero, the additive biases are cancelling sums, and the user encoder's additive attention sees nearly
    uniform pooling weights (its gradients are 1e-13: below what an fp16 forward leaves of them)."""
    return (name.endswith("W_K.bias") or name.endswith("additive_attention.linear.bias")
            or name.startswith("user_encoder.additive_attention."))


def test_fp16_mode_properties_at_full_size(full):
    """The benchmarked mode at the benchmarked size: run-to-run determinism, user-permutation equivariance (a user's
    scores do not depend on its batch position: bit-exact), distance to the exact fp32 scores inside north_star's bar,
    the backward linear in d(scores) (a factor 2 is exact in fp16 too), padding-row and untouched-row gradients zero,
    and the compact path against the dense path."""
    from tests.test_hip_parity import make_model
    shape, _, batch, tb = full
    model = make_model(shape, synth.make_params(shape, seed=0))      # fresh weights (the shared fixture has been trained on)
    eng, flat, lay = model.engine, model._flat, model._layout
    ref = _scores(model, tb, training=False).clone()
    valid = tb["candidate_mask"] == 1
    try:
        model.config.precision = "fp16"
        eng = model.engine
        assert eng.precision == "fp16"
        s1 = _scores(model, tb).clone()
        assert torch.equal(s1, _scores(model, tb))
        e = (s1 - ref)[valid].abs().double()
        rms, p999, err = float((e * e).mean().sqrt()), float(torch.quantile(e, 0.999)), float(e.max())
        print("full size fp16 vs fp32 over %d scores of rms %.3f: rms %.2e, 99.9 %% %.2e, max %.2e" % (
            e.numel(), float((ref[valid].double() ** 2).mean().sqrt()), rms, p999, err))
        # north_star's 1e-4 is asserted against the REFERENCE on fixture g2 (20 scores, max 4.4e-5); over 2 555 scores the
        # same error distribution (rms 2.8e-5 = 5e-4 of the score scale: one fp16 rounding) has its maximum AT the bar
        assert 1e-7 < rms < 4e-5 and p999 < 1e-4 and err < 1.5e-4
        perm = torch.from_numpy(np.random.default_rng(4).permutation(shape.batch_size)).cuda()
        assert torch.equal(_scores(model, {k: v[perm] for k, v in tb.items()}), s1[perm])
        d1 = (torch.randn(s1.shape, generator=torch.Generator().manual_seed(5)) * 1e-3).cuda()
        grads = []
        for scale in (1.0, 2.0):
            _scores(model, tb)
            gf = torch.zeros_like(flat)
            eng.backward(flat, gf, d1 * scale)
            grads.append(gf)
        for name in lay.names:
            if noise_only(name):
                continue
            a, b = lay.view(grads[0], name), lay.view(grads[1], name)
            # (a factor 2 is exact in fp16 except for values in the subnormal range -- the tokens with tiny pooling weights --
            # which keep fewer bits than their doubles: the bound is the mode's gradient tolerance, GRAD_REL of test_hip_fp16)
            assert float((2 * a - b).abs().max()) <= 4e-3 * float(a.abs().max()) + 1e-12, name
        emb = lay.view(grads[0], "news_encoder.word_embedding.0.weight")
        ids = torch.cat([tb["browsed_titles"].reshape(-1), tb["candidate_titles"].reshape(-1)])
        untouched = torch.ones(shape.n_words, dtype=torch.bool, device="cuda")
        untouched[ids.unique()] = False
        assert not emb[0].any() and not emb[untouched].any()
        # compact (padding tokens skipped) against dense, same mode
        eng.pad_row_zero = False
        s_dense = _scores(model, tb).clone()
        g_dense = torch.zeros_like(flat)
        eng.backward(flat, g_dense, d1)
        eng.pad_row_zero = model._pad_zero
        assert float((s_dense - s1)[valid].abs().max()) < 2e-5
        for name in lay.names:
            if noise_only(name):
                continue
            a, b = lay.view(grads[0], name), lay.view(g_dense, name)
            assert float((a - b).abs().max()) <= 4e-3 * float(b.abs().max()) + 1e-12, name
    finally:
        model.config.precision = "fp32"
        model.engine.pad_row_zero = model._pad_zero
