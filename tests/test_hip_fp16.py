"""precision = "fp16": the fused one-wavefront-per-sequence kernels (csrc/fused16.hip) against the reference
fixtures and the oracle.  fp16 operands carry 11 significant bits, so the bar here is north_star's own
(scores within 1e-4 of the reference), not the 1e-5 of the fp32 / bf16x3 modes; the measured errors are printed.

The mode's default keeps the user encoder (3 % of the flops) in bf16x3; `fp16_user` = True puts it on the fused fp16
kernels as well (config.fp16_user_encoder) -- that variant is tested against a 1.5e-4 bar (it sits AT 1e-4 on large
batches, which is why it is not the default)."""
import os

import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model, tbatch

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4          # north_star: click scores within 1e-4 of the reference
USER16 = [False, True]    # config.fp16_user_encoder


def score_terms(aux, valid=None):
    """What a score is made of: sum_f |cand_f user_f| per (user, candidate).  fp16 rounding is RELATIVE to these terms, not to
    the score they sum to: MIND-shaped models (d = 300, H = 50, L = 30) at initialisation have terms of 0.6 ... 0.9 for scores
    of rms 0.05; narrow synthetic shapes (d = 60, one-token titles) build a score of 0.2 from terms of 8."""
    cand, user = np.asarray(aux["cand"]), np.asarray(aux["user"])
    t = np.abs(cand * user[:, None, :]).sum(-1)
    return float(t[valid].max() if valid is not None else t.max())


def score_bar(o_scores, fp16_user=False, terms=None):
    """north_star's bar is ABSOLUTE: every score within 1e-4 of the reference's.  It is asserted as such -- no scaling with the
    score -- wherever the scores are in the range MIND-shaped models produce (|score| <= 0.4; v0 at initialisation <= 0.2,
    nrms_v1 <= 0.36), i.e. in every test of the benchmarked geometry and of the reference fixtures.
    FINDING of round 4 (the scaled bar of round 3 hid it): the fp16 mode's error is relative to the TERMS of a score,
    sum_f |cand_f user_f| (score_terms) -- measured 8.5e-5 per unit of terms at d = 300 (7.6e-5 at the bench size, terms 0.9)
    and up to 2.2e-4 per unit in narrow synthetic shapes (d = 60 ... 80, one-word titles, one-slot histories: no averaging over
    tokens).  So the absolute 1e-4 is a property of MIND-shaped inputs, not of every shape the kernels accept: the shape fuzz
    and the odd-shape tests, whose terms reach 3 ... 18, pass `terms` and are held to 3e-4 of them instead (kernel indexing is
    what they test), never below 1e-4.  `fp16_user`: the documented non-default variant (user encoder in fp16 too), 1.5 x."""
    k = 1.5 if fp16_user else 1.0
    if terms is not None:
        # (the fp16 user encoder adds its own four roundings: 6e-4 of the terms -- measured 4.8e-4 on the 8-wide `tiny` shape)
        return max(k * SCORE_TOL, (6e-4 if fp16_user else 3e-4) * terms)
    return k * SCORE_TOL * max(1.0, float(np.abs(o_scores).max()) / 0.4)
VEC_TOL = 1.5e-3          # news / user vectors (norm ~ 1..10): 2^-11 relative per rounding


def _padded_to_model_cols(keep_padded, n_heads, dk):
    """keep mask over the padded [.., 320] context layout (32 columns per head) -> the model's [.., n_heads * dk]."""
    k = keep_padded.reshape(keep_padded.shape[0], 10, 32)[:, :n_heads, :dk]
    return k.reshape(keep_padded.shape[0], n_heads * dk)


@pytest.mark.parametrize("fp16_user", USER16)
def test_fp16_scores_within_bar_of_reference_fixture(golden_dir, fp16_user):
    g = np.load(os.path.join(golden_dir, "g2_mind.npz"), allow_pickle=False)
    shape = synth.G2_MIND
    params = synth.make_params(shape, seed=21)
    batch = synth.make_batch(shape, seed=22, ragged=True)
    model = make_model(shape, params, precision="fp16", fp16_user=fp16_user).eval()
    B, H, L = batch["browsed_titles"].shape
    with torch.no_grad():
        for dedup in (True, False):
            model.dedup_inference = dedup
            s = model(tbatch(batch)).cpu().numpy()
            err = float(np.abs(s - g["scores"]).max())
            print("fp16 g2 (dedup=%s): max |score - reference| = %.3e" % (dedup, err))
            assert err < SCORE_TOL
        hist = model.get_news_vector(torch.from_numpy(batch["browsed_titles"]).reshape(B * H, L)).view(B, H, -1)
    verr = float(np.abs(hist.cpu().numpy() - g["hist"]).max())
    print("fp16 g2: max |news vector - reference| = %.3e (scale %.2f)" % (verr, float(np.abs(g["hist"]).max())))
    assert verr < VEC_TOL


@pytest.mark.parametrize("fp16_user", USER16)
@pytest.mark.parametrize("case", ["g1_odd", "tiny", "bench_small", "all_padding", "nonzero_pad_row"])
def test_fp16_forward_shapes_against_oracle(case, fp16_user):
    """Both encoders through the fused kernel (histories of at most 32 slots) on awkward shapes: d_k = 6 with 10
    heads (KP = 64, DP = 320, QP = 32), minimum sizes, all-padding titles (closed form), an empty-history user,
    masked candidates, a table whose padding row is not zero (dense path)."""
    from oracle import nrms_oracle as orc
    kw = dict(seed=102, ragged=True, min_title=1, empty_history_user=True, all_pad_title=True, mask_some_candidates=True)
    pad_zero = True
    if case == "g1_odd":
        shape = synth.G1_ODD
    elif case == "tiny":
        shape = synth.Shape(n_words=64, word_embed_size=8, num_attention_heads=2, query_vector_dim=4, batch_size=1,
                            history_len=1, n_candidates=1, n_words_title=1)
        kw = dict(seed=102, ragged=True, min_title=1)
    elif case == "bench_small":
        shape = synth.Shape(n_words=1000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                            batch_size=9, history_len=32, n_candidates=5, n_words_title=30)
    elif case == "all_padding":
        shape = synth.Shape(n_words=300, word_embed_size=60, num_attention_heads=6, query_vector_dim=32, batch_size=5,
                            history_len=9, n_candidates=4, n_words_title=11)
    else:
        shape = synth.Shape(n_words=300, word_embed_size=120, num_attention_heads=6, query_vector_dim=64, batch_size=12,
                            history_len=20, n_candidates=4, n_words_title=17)
        pad_zero = False
    params = synth.make_params(shape, seed=101, pad_row_zero=pad_zero)
    batch = synth.make_batch(shape, **kw)
    if case == "all_padding":
        batch["browsed_titles"][:] = 0
        batch["candidate_titles"][:] = 0
    model = make_model(shape, params, precision="fp16", fp16_user=fp16_user).eval()
    model.dedup_inference = False
    with torch.no_grad():
        s = model(tbatch(batch)).cpu().numpy()
    assert model.engine.pad_row_zero is pad_zero and model.engine.fp16_user_encoder is fp16_user
    p = orc.to_torch(params)
    with torch.no_grad():
        o_scores, aux = orc.forward(p, batch, shape.num_attention_heads)
    o_scores = o_scores.numpy()
    valid = batch["candidate_mask"] == 1
    err = float(np.abs(s - o_scores)[valid].max())
    print("fp16 %s (user fp16=%s): max |score - oracle| = %.3e (score scale %.3f)" % (case, fp16_user, err, float(np.abs(o_scores[valid]).max())))
    terms = score_terms(aux, valid) if case not in ("bench_small",) else None     # MIND-shaped: the absolute bar
    assert err < score_bar(o_scores[valid], fp16_user, terms), (err, terms)
    assert (s[~valid] == np.float32(-1e9)).all()
    # the encoders on their own
    B, H, L = batch["browsed_titles"].shape
    with torch.no_grad():
        nv = model.get_news_vector(torch.from_numpy(batch["browsed_titles"]).reshape(B * H, L)).view(B, H, -1)
        uv = model.get_user_vector(torch.from_numpy(aux["hist"].numpy()))
    assert float(np.abs(nv.cpu().numpy() - aux["hist"].numpy()).max()) < VEC_TOL
    assert float(np.abs(uv.cpu().numpy() - aux["user"].numpy()).max()) < VEC_TOL


def test_fp16_train_mode_forward_replays_its_dropout_masks():
    """Dropout on, no backward: the fused kernel's masks (embedding dropout: the fp32 path's counters; context
    dropout: counters over the padded [tokens, 32 n_heads] layout) exported and replayed through the oracle."""
    from oracle import nrms_oracle as orc
    shape = synth.Shape(n_words=500, word_embed_size=60, num_attention_heads=6, query_vector_dim=32,
                        batch_size=6, history_len=9, n_candidates=4, n_words_title=12)
    params = synth.make_params(shape, seed=5)
    batch = synth.make_batch(shape, seed=6, ragged=True, min_title=2, all_pad_title=True)
    model = make_model(shape, params, dropout=0.2, precision="fp16").train()
    tb = {k: v.cuda() for k, v in tbatch(batch).items()}
    eng = model.engine
    seed = 0x1234567
    with torch.no_grad():
        s = eng.forward(model._flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=False,
                        p_drop=0.2, seed=seed).cpu().numpy()
    n_titles = shape.batch_size * (shape.history_len + shape.n_candidates)
    L, d, h = shape.n_words_title, shape.word_embed_size, shape.num_attention_heads
    ke = eng.dropout_keep_mask(seed, 0, n_titles * L, 0.2).cpu().view(n_titles, L, d)
    kc_pad = eng.dropout_keep_mask(seed, 1, n_titles * L, 0.2, fp16_ctx=True).cpu().numpy()
    kc = torch.from_numpy(_padded_to_model_cols(kc_pad, h, d // h)).view(n_titles, L, d)
    assert 0.77 < float(kc.float().mean()) < 0.83
    pt = orc.to_torch(params)
    with torch.no_grad():
        o_scores, _ = orc.forward(pt, batch, h, p_drop=0.2, keep={"embed": ke, "ctx": kc})
    err = float(np.abs(s - o_scores.numpy()).max())
    print("fp16 dropout replay: max |score - oracle| = %.3e" % err)
    assert err < score_bar(o_scores.numpy())


def test_fp16_mode_rejects_shapes_outside_the_fused_kernels():
    """The C ABI refuses what the fused kernels do not cover (the engine routes such encoder passes to bf16x3)."""
    import ctypes as C
    from pytorch_news_recommender_amd import _lib
    lib = _lib.load()
    for kw in (dict(d_model=320), dict(d_model=384, n_heads=12), dict(n_heads=5), dict(q_dim=256), dict(use_output_proj=1),
               dict(mask_mode=1)):
        f = dict(n_seq=4, seq_len=30, d_model=300, n_heads=10, q_dim=200, vocab=0, p_drop_embed=0.0, p_drop_ctx=0.0,
                 precision=_lib.NRMS_PRECISION_FP16, use_output_proj=0, mask_mode=0, flags=0, seed=0)
        f.update(kw)
        desc = _lib.EncoderDesc(**f)
        assert lib.nrms_encoder_fwd_scratch_bytes(C.byref(desc)) == 0
        assert b"fp16" in lib.nrms_last_error()


# ---- training: the fused fp16 backward (csrc/fused16_bwd.hip) -----------------------------------------------------
# fp16 gradient tensors carry 11 significant bits (x a power-of-two loss scale), so gradients are compared
# relative to each tensor's scale: max |got - ref| <= GRAD_REL * max |ref| + GRAD_ABS.
GRAD_REL, GRAD_ABS = 4e-3, 2e-6


def _grad_report(grads, ref, names, tag, abs_floor=GRAD_ABS):
    worst = 0.0
    for n in names:
        r = np.asarray(ref[n])
        scale = float(np.abs(r).max())
        err = float(np.abs(np.asarray(grads[n]) - r).max())
        rel = err / (scale + 1e-30)
        worst = max(worst, rel if scale > 1e-7 else 0.0)
        print("   %-8s %-62s scale %.2e  max err %.2e  (%.1e of scale)" % (tag, n, scale, err, rel))
        floor = abs_floor
        if n.endswith("W_K.bias") and n.replace("W_K", "W_Q") in ref:
            # analytically zero (softmax is invariant to a per-query constant): what is measured is the rounding noise of
            # the dK terms that cancel, which scale like d(W_Q.bias)
            floor = max(floor, GRAD_REL * float(np.abs(np.asarray(ref[n.replace("W_K", "W_Q")])).max()))
        assert err <= GRAD_REL * scale + floor, (tag, n, err, scale)
    return worst


def test_fp16_gradients_against_reference_fixture(golden_dir):
    from tests.test_hip_parity import fwd_bwd
    g = np.load(os.path.join(golden_dir, "g2_mind.npz"), allow_pickle=False)
    shape = synth.G2_MIND
    params = synth.make_params(shape, seed=21)
    batch = synth.make_batch(shape, seed=22, ragged=True)
    model = make_model(shape, params, precision="fp16").train()          # dropout 0
    scores, loss, grads = fwd_bwd(model, batch)
    assert float(np.abs(scores - g["scores"]).max()) < SCORE_TOL and abs(loss - float(g["loss"])) < SCORE_TOL
    emb = "news_encoder.word_embedding.0.weight"
    ref = {n: g["grad/" + n] for n in synth.param_names() if n != emb}
    _grad_report(grads, ref, list(ref), "g2")
    rows = g["rows"]
    _grad_report({emb: grads[emb][rows]}, {emb: g["grad_rows/" + emb]}, [emb], "g2 rows")
    np.testing.assert_allclose(grads[emb].sum(axis=1), g["grad_rowsum/" + emb], rtol=5e-3, atol=2e-4)
    assert not grads[emb][0].any()


@pytest.mark.parametrize("fp16_user", USER16)
@pytest.mark.parametrize("case", ["g1_odd", "bench_small", "all_padding", "nonzero_pad_row", "dropout", "dropout_bench"])
def test_fp16_forward_backward_against_oracle(case, fp16_user):
    """`dropout_bench`: the benchmarked geometry -- d = 300 (KP = DP = 320), 10 heads x 30 (the 16-bit-field mask over
    10 x 32 padded columns), q = 200 (QP = 224), 30-word titles (SB = 1 kernels), 50-slot histories (the SB = 2 user-encoder
    kernels when fp16_user), dropout 0.2 on both sites, padding-skipping compact path -- replayed through the oracle with
    the kernels' own keep masks (nrms_v0.py:137,171-173 semantics): scores and all 19 gradients."""
    from oracle import nrms_oracle as orc
    from tests.test_hip_parity import fwd_bwd
    kw = dict(seed=102, ragged=True, min_title=1, empty_history_user=True, all_pad_title=True, mask_some_candidates=True)
    pad_zero, p_drop = True, 0.0
    if case == "g1_odd":
        shape = synth.G1_ODD
    elif case == "bench_small":
        shape = synth.Shape(n_words=1000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                            batch_size=9, history_len=32, n_candidates=5, n_words_title=30)
    elif case == "all_padding":
        shape = synth.Shape(n_words=300, word_embed_size=60, num_attention_heads=6, query_vector_dim=32, batch_size=5,
                            history_len=9, n_candidates=4, n_words_title=11)
    elif case == "nonzero_pad_row":
        shape = synth.Shape(n_words=300, word_embed_size=120, num_attention_heads=6, query_vector_dim=64, batch_size=12,
                            history_len=20, n_candidates=4, n_words_title=17)
        pad_zero = False
    elif case == "dropout":
        shape = synth.Shape(n_words=500, word_embed_size=60, num_attention_heads=6, query_vector_dim=32,
                            batch_size=6, history_len=9, n_candidates=4, n_words_title=12)
        p_drop = 0.2
    else:
        shape = synth.Shape(n_words=2000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                            batch_size=6, history_len=50, n_candidates=5, n_words_title=30)
        p_drop = 0.2
    params = synth.make_params(shape, seed=101, pad_row_zero=pad_zero)
    batch = synth.make_batch(shape, **kw)
    if case == "all_padding":
        batch["browsed_titles"][:] = 0
        batch["candidate_titles"][:] = 0
    model = make_model(shape, params, dropout=p_drop, precision="fp16", fp16_user=fp16_user).train()
    scores, loss, grads = fwd_bwd(model, batch)
    assert model.engine.pad_row_zero is pad_zero
    keep = None
    if p_drop > 0:
        sv = model.engine._saved
        n_titles = shape.batch_size * (shape.history_len + shape.n_candidates)
        L, d, h = shape.n_words_title, shape.word_embed_size, shape.num_attention_heads
        ke = model.engine.dropout_keep_mask(sv["seed"], 0, n_titles * L, p_drop).cpu().view(n_titles, L, d)
        kc = model.engine.dropout_keep_mask(sv["seed"], 1, n_titles * L, p_drop, fp16_ctx=True).cpu().numpy()
        keep = {"embed": ke, "ctx": torch.from_numpy(_padded_to_model_cols(kc, h, d // h)).view(n_titles, L, d)}
    o_scores, o_loss, o_grads, aux = orc.loss_and_grads(params, batch, shape.num_attention_heads, p_drop=p_drop, keep=keep)
    valid = batch["candidate_mask"] == 1
    err = float(np.abs(scores - o_scores)[valid].max())
    print("fp16 train %s (user fp16=%s): max |score - oracle| = %.3e, |loss diff| %.2e" % (case, fp16_user, err, abs(loss - o_loss)))
    terms = None if case in ("dropout_bench", "bench_small") else score_terms(aux, valid)      # MIND-shaped: the absolute bar
    assert err < score_bar(o_scores[valid], fp16_user, terms), (err, terms)
    # all_padding: every title is the same vector, the true bias gradients are differences of equal terms (~1e-8)
    # while each term is ~0.1: what is left is the fp16 rounding of the terms, ~1e-4 absolute
    _grad_report(grads, o_grads, synth.param_names(), case, abs_floor=2e-4 if case == "all_padding" else GRAD_ABS)
    assert not grads["news_encoder.word_embedding.0.weight"][0].any()


def test_fp16_sum_reduced_loss_through_autograd():
    """A loss the fixed 128 x batch scale of round 2 could not carry: CrossEntropyLoss(reduction="sum") at 512 users makes
    |d(scores)| O(1) instead of O(1/512) -- x 65536 that overflowed fp16 and sent NaNs through Adam.  The loss scale is
    now derived on the device from max |dout| of each backward call, so any reduction / batch / world size works: the
    gradients must be finite and equal batch x the oracle's mean-loss gradients."""
    from oracle import nrms_oracle as orc
    # (a vocabulary large enough that no word occurs more than 64 times in the step: the grouped scatter sums longer buckets
    #  in an order that is reproducible only chunk by chunk, csrc/embed.hip)
    shape = synth.Shape(n_words=30000, word_embed_size=60, num_attention_heads=6, query_vector_dim=32,
                        batch_size=512, history_len=9, n_candidates=4, n_words_title=12)
    params = synth.make_params(shape, seed=7)
    batch = synth.make_batch(shape, seed=8, ragged=True, min_title=1, mask_some_candidates=True)
    res = {}
    for red in ("sum", "mean"):
        model = make_model(shape, params, precision="fp16", fp16_user=True).train()
        model.zero_grad()
        scores = model(tbatch(batch))
        y = torch.zeros(len(scores), dtype=torch.long, device=scores.device)
        torch.nn.CrossEntropyLoss(reduction=red)(scores, y).backward()
        res[red] = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}
        assert all(np.isfinite(g).all() for g in res[red].values()), red
    _, _, o_grads, _ = orc.loss_and_grads(params, batch, shape.num_attention_heads)
    B = shape.batch_size
    _grad_report(res["mean"], o_grads, synth.param_names(), "mean")
    _grad_report({n: g / B for n, g in res["sum"].items()}, o_grads, synth.param_names(), "sum/B")
    for n in synth.param_names():          # a power-of-two change of scale: the two runs carry the same fp16 values
        np.testing.assert_array_equal(res["sum"][n], res["mean"][n] * np.float32(B), err_msg=n)


def test_fp16_fused_train_steps_track_the_reference(golden_dir):
    """Model.train_step x3 in fp16 mode against the reference stepped by torch.optim.Adam (fixture g5): losses to the
    score bar; parameters move by at most lr per step, so fp16 gradient noise shows up as a fraction of 1e-3."""
    from tests.test_hip_parity import ILL_CONDITIONED
    g = np.load(os.path.join(golden_dir, "g5_adam.npz"), allow_pickle=False)
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=51)
    model = make_model(shape, params, precision="fp16").train()
    model.config.learning_rate = 1e-3
    losses = []
    for t in range(3):
        batch = synth.make_batch(shape, seed=52 + t, ragged=True, min_title=1)
        losses.append(float(model.train_step(tbatch(batch))) / shape.batch_size)
    np.testing.assert_allclose(losses, g["losses"], atol=2e-4)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for n in synth.param_names():
        diff = np.abs(sd[n] - g["param/" + n])
        print("   fp16 adam %-62s max %.2e  median %.2e" % (n, float(diff.max()), float(np.median(diff))))
        # Adam moves a parameter by ~lr per step whatever the gradient's size: where a gradient element is at the fp16
        # noise level its SIGN is noise, and the element may end up to 3 lr away; the bulk must agree closely
        assert diff.max() < 3.1e-3, n
        if not n.endswith(ILL_CONDITIONED):
            assert np.median(diff) < 2e-5 and float((diff > 2e-4).mean()) < 0.03, n


def _finite(*ts):
    return all(bool(torch.isfinite(t).all()) for t in ts)


def test_fp16_gradient_overflow_is_counted_skipped_and_backed_off():
    """the loss scale is picked from max |dout| alone; weights of unusual norm can push a derived fp16 tensor
    (dS, dQKV, dX) past 65504, and an inf would travel into the table / weight gradients and into Adam's m and v for good.
    Now: the guarded optimizer leaves non-finite elements out and counts them on the device, the count reaches the host
    asynchronously, and the following steps run with more head room (desc.loss_scale = -n).
    (a) forced: a fixed loss scale of 2^40 makes every fp16 gradient inf -- nothing becomes non-finite, the step is reported;
    (b) natural: the news encoder's W_Q, W_K, W_V scaled by 3e3 overflow the default head room in dX; the back-off finds a scale
        that fits within a few steps, parameters and moments stay finite throughout, and the gradients at that scale agree
        with the exact fp32 mode's."""
    import warnings
    shape = synth.Shape(n_words=1000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=16, history_len=50, n_candidates=5, n_words_title=30)
    params = synth.make_params(shape, seed=5)
    tb = {k: torch.from_numpy(v).cuda() for k, v in synth.make_batch(shape, seed=6).items()}
    model = make_model(shape, params, precision="fp16").train()
    model.config.learning_rate = 1e-3
    eng = model.engine
    model.train_step(tb)
    assert eng.poll_grad_overflow(block=True) == 0 and eng.grad_overflow_steps == 0
    st = model._opt
    wq = "news_encoder.multihead_self_attention.W_Q.weight"
    before = model._layout.view(model._flat, wq).clone()
    user_before = model._layout.view(model._flat, "user_encoder.multihead_self_attention.W_Q.weight").clone()
    eng.loss_scale_override = 2.0 ** 40
    model.train_step(tb)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert eng.poll_grad_overflow(block=True) == 2
    assert eng.grad_overflow_steps == 1 and any("non-finite" in str(w.message) for w in rec)
    assert _finite(model._flat, st["m"], st["v"])
    assert torch.equal(model._layout.view(model._flat, wq), before)          # its gradient was inf everywhere: left alone
    assert not torch.equal(model._layout.view(model._flat, "user_encoder.multihead_self_attention.W_Q.weight"), user_before)
    eng.loss_scale_override = None
    eng.loss_scale_backoff = 0
    # the autograd path (caller's own optimizer): non-finite elements become zeros, counted
    eng.loss_scale_override = 2.0 ** 40
    model.zero_grad()
    s = model({k: v for k, v in tb.items()})
    torch.nn.functional.cross_entropy(s, torch.zeros(len(s), dtype=torch.long, device=s.device)).backward()
    assert all(_finite(p.grad) for p in model.parameters())
    assert not model.news_encoder.multihead_self_attention.W_Q.weight.grad.any()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert eng.poll_grad_overflow(block=True) == 2 and eng.grad_overflow_steps == 2
    eng.loss_scale_override = None
    eng.loss_scale_backoff = 0

    # (b) natural overflow: W_Q, W_K, W_V of the news encoder scaled by 3e3.  The attention becomes one-hot (fp32 softmax: fine),
    # dV = P^T d(ctx) stays O(100) at the default scale, and dX = dQKV W' -- carried as loss-scaled fp16 -- reaches ~6e5 > 65504.
    # The upstream gradient is given directly (with weights like these the CE gradient can be exactly zero).
    wv = "news_encoder.multihead_self_attention.W_V.weight"
    p2 = dict(params)
    for nm in ("W_Q", "W_K", "W_V"):
        p2["news_encoder.multihead_self_attention.%s.weight" % nm] = params["news_encoder.multihead_self_attention.%s.weight" % nm] * 3e3
    m2 = make_model(shape, p2, precision="fp16").train()
    e2, f2 = m2.engine, m2._flat
    dsc = (torch.randn(shape.batch_size, shape.n_candidates, generator=torch.Generator().manual_seed(8)) * 1e-3).cuda()
    mom, var = torch.zeros_like(f2), torch.zeros_like(f2)
    seen, clean_at = 0, None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(14):
            e2.forward(f2, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=True)
            g2 = torch.zeros_like(f2)
            e2.backward(f2, g2, dsc)
            e2.adam_step(f2, g2, mom, var, i + 1, lr=1e-7)          # guarded: precision fp16
            e2.note_grad_check()
            e2.poll_grad_overflow(block=True)
            assert _finite(f2, mom, var), i
            if e2.grad_overflow_steps == seen and seen > 0:
                clean_at = i
                break
            seen = e2.grad_overflow_steps
    print("natural overflow: %d overflowing steps, clean at step %s with back-off 2^-%d" % (seen, clean_at, e2.loss_scale_backoff))
    assert seen >= 1 and clean_at is not None
    # (c) what a back-off costs: ORDINARY weights with six more powers of two of head room (max |dout| into [1, 2)) against the
    # exact fp32 mode -- the fp16 tensors have 2^20 below the default target before they flush, so the gradients still meet the
    # mode's bar.  (At the weights of (b) itself such a comparison says nothing: news vectors of norm 4 000 give the user
    # encoder logits of 1e7, its attention is one-hot and flips with the last bit of ANY arithmetic.)
    m3 = make_model(shape, params, precision="fp16").train()
    g = {}
    for prec, backoff in (("fp16", 6), ("fp32", 0)):
        m3.config.precision = prec
        eng3 = m3.engine
        eng3.loss_scale_backoff = backoff
        sc = eng3.forward(m3._flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=True)
        _, dce = eng3.ce_loss(sc, grad_scale=1.0 / shape.batch_size)
        g[prec] = torch.zeros_like(m3._flat)
        eng3.backward(m3._flat, g[prec], dce)
    assert _finite(g["fp16"])
    for name in (wv, "news_encoder.word_embedding.0.weight", "news_encoder.multihead_self_attention.W_Q.weight",
                 "news_encoder.additive_attention.linear.weight"):
        a, b = m3._layout.view(g["fp16"], name).double(), m3._layout.view(g["fp32"], name).double()
        rel = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-300)
        print("   fp16 with 2^-6 head room vs fp32 grad %-60s scale %.2e  max err %.1e of scale" % (name, float(b.abs().max()), rel))
        # (the additive weight's gradient is 1e-6 at initialisation -- 3e-3 ... 4e-3 of its scale at the default head room)
        assert rel < (1.5e-2 if "additive_attention" in name else 6e-3), (name, rel)
