"""precision = "fp16" for nrms_v1's news encoder (model/nrms_v1.py:109-162: heads of 50 columns, W_O, dropout after W_O):
csrc/fused16_v1.hip against the oracle, at the fp16 mode's bar (north_star: scores within 1e-4 of the reference)."""
import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import _lib, synth
from tests.test_hip_v1 import make_v1

pytestmark = pytest.mark.gpu

V1_SHAPES = {
    # res_logs.md:4 / config.py:87-88: 300-wide embeddings, six title heads of 50, ten user heads, 20-word titles
    "reference": (synth.Shape(n_words=5000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                              batch_size=16, history_len=50, n_candidates=5, n_words_title=20), 6, 1),
    # 30-word titles: long titles (more than 15 words) next to paired short ones
    "long_titles": (synth.Shape(n_words=3000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                                batch_size=8, history_len=20, n_candidates=5, n_words_title=30), 6, 5),
    # heads of 40: no leftover features
    "dk40": (synth.Shape(n_words=900, word_embed_size=240, num_attention_heads=6, query_vector_dim=64,
                         batch_size=5, history_len=11, n_candidates=3, n_words_title=17), 6, 2),
    # four heads of 50 on 200 columns, an odd title length
    "dk50_h4": (synth.Shape(n_words=700, word_embed_size=200, num_attention_heads=4, query_vector_dim=100,
                            batch_size=4, history_len=9, n_candidates=4, n_words_title=13), 4, 1),
}


# The opt-in fused fp16 news encoder of nrms_v1 at 512 users (measured max 1.56e-4): NOT inside north_star's 1e-4, stated as such
V1_FP16_OPT_IN_BAR = 2e-4


def v1_bar(o_scores):
    return V1_FP16_OPT_IN_BAR * max(1.0, float(np.abs(o_scores).max()) / 0.4)


def _news_precision(model, n, L, training):
    return model.engine._desc("news_encoder", n, L, training=training).precision


@pytest.mark.parametrize("case", sorted(V1_SHAPES))
def test_v1_fp16_forward_against_oracle(case):
    from oracle import nrms_oracle as orc
    shape, title_heads, min_title = V1_SHAPES[case]
    params = synth.make_params_v1(shape, seed=31)
    batch = synth.make_batch(shape, seed=32, ragged=True, min_title=min_title, all_pad_title=True, mask_some_candidates=True)
    model = make_v1(shape, params, title_heads, precision="fp16").eval()
    assert _news_precision(model, 8, shape.n_words_title, False) == _lib.NRMS_PRECISION_FP16     # not the bf16x3 fallback
    with torch.no_grad():
        scores = model({k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}).cpu().numpy()
        nv = model.get_news_vector(torch.from_numpy(batch["browsed_titles"]).reshape(-1, shape.n_words_title)).cpu().numpy()
    pt = orc.to_torch(orc.v1_to_v0_names(params))
    with torch.no_grad():
        o_scores, _ = orc.forward(pt, batch, shape.num_attention_heads, news_heads=title_heads, embed_dropout=False)
        o_nv = orc.news_encoder(pt, torch.from_numpy(batch["browsed_titles"]).reshape(-1, shape.n_words_title), title_heads,
                                embed_dropout=False).numpy()
    verr = float(np.abs(nv - o_nv).max())
    err = float(np.abs(scores - o_scores.numpy()).max())
    print("v1 fp16 %s: max |news vector - oracle| = %.3e (max |v| %.2f), max |score - oracle| = %.3e" %
          (case, verr, float(np.abs(o_nv).max()), err))
    assert verr < 1.5e-3 * max(1.0, float(np.abs(o_nv).max()))
    assert err < v1_bar(o_scores.numpy())


def test_v1_fp16_forward_replays_its_dropout_mask():
    """Dropout after W_O (nrms_v1.py:161): the kernel's keep mask lives on the padded [tokens, 10 x 32] layout of the
    projection's output blocks (d / 10 columns each)."""
    from oracle import nrms_oracle as orc
    shape, title_heads, _ = V1_SHAPES["reference"]
    params = synth.make_params_v1(shape, seed=41)
    batch = synth.make_batch(shape, seed=42, ragged=True, min_title=1, all_pad_title=True)
    model = make_v1(shape, params, title_heads, dropout=0.2, precision="fp16").train()
    eng = model.engine
    tb = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in batch.items()}
    seed = 0x7654321
    with torch.no_grad():
        s = eng.forward(model._flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=False,
                        p_drop=0.2, seed=seed).cpu().numpy()
    assert _news_precision(model, 8, shape.n_words_title, False) == _lib.NRMS_PRECISION_FP16
    n_titles = shape.batch_size * (shape.history_len + shape.n_candidates)
    L, d = shape.n_words_title, shape.word_embed_size
    kc_pad = eng.dropout_keep_mask(seed, 1, n_titles * L, 0.2, fp16_ctx=True).cpu().numpy()
    kc = kc_pad.reshape(-1, 10, 32)[:, :, :d // 10].reshape(n_titles, L, d)
    pt = orc.to_torch(orc.v1_to_v0_names(params))
    with torch.no_grad():
        o_scores, _ = orc.forward(pt, batch, shape.num_attention_heads, p_drop=0.2, keep={"ctx": torch.from_numpy(kc)},
                                  news_heads=title_heads, embed_dropout=False)
    err = float(np.abs(s - o_scores.numpy()).max())
    print("v1 fp16 dropout replay: max |score - oracle| = %.3e" % err)
    assert err < v1_bar(o_scores.numpy())


# ---- training: csrc/fused16_v1_bwd.hip --------------------------------------------------------------------------------
GRAD_REL, GRAD_ABS = 4e-3, 2e-6           # as tests/test_hip_fp16.py: relative to each tensor's scale


def _v1_grad_report(grads, o_grads, back, tag):
    for v0name, r in o_grads.items():
        r = np.asarray(r)
        got = np.asarray(grads[back[v0name]])
        scale, err = float(np.abs(r).max()), float(np.abs(got - r).max())
        print("   %-10s %-62s scale %.2e  max err %.2e  (%.1e of scale)" % (tag, v0name, scale, err, err / (scale + 1e-30)))
        floor = GRAD_ABS
        if "news_encoder" in v0name and v0name.endswith("W_K.bias"):
            # analytically zero: the rounding noise of cancelling dK terms, which scale like d(W_Q.bias)
            floor = max(floor, GRAD_REL * float(np.abs(np.asarray(o_grads[v0name.replace("W_K", "W_Q")])).max()))
        assert err <= GRAD_REL * scale + floor, (tag, v0name, err, scale)


@pytest.mark.parametrize("p_drop", [0.0, 0.2])
@pytest.mark.parametrize("case", sorted(V1_SHAPES))
def test_v1_fp16_forward_backward_against_oracle(case, p_drop):
    from oracle import nrms_oracle as orc
    from tests.test_hip_v1 import fwd_bwd
    shape, title_heads, min_title = V1_SHAPES[case]
    params = synth.make_params_v1(shape, seed=51)
    batch = synth.make_batch(shape, seed=52, ragged=True, min_title=min_title, all_pad_title=True, mask_some_candidates=True)
    model = make_v1(shape, params, title_heads, dropout=p_drop, precision="fp16").train()
    assert _news_precision(model, 8, shape.n_words_title, True) == _lib.NRMS_PRECISION_FP16      # the fused backward, not bf16x3
    scores, loss, grads = fwd_bwd(model, batch)
    keep = None
    if p_drop > 0:
        sv = model.engine._saved
        n_titles = shape.batch_size * (shape.history_len + shape.n_candidates)
        L, d = shape.n_words_title, shape.word_embed_size
        kc = model.engine.dropout_keep_mask(sv["seed"], 1, n_titles * L, p_drop, fp16_ctx=True).cpu().numpy()
        keep = {"ctx": torch.from_numpy(kc.reshape(-1, 10, 32)[:, :, :d // 10].reshape(n_titles, L, d).copy())}
    v0 = orc.v1_to_v0_names(params)
    o_scores, o_loss, o_grads, _ = orc.loss_and_grads(v0, batch, shape.num_attention_heads, p_drop=p_drop, keep=keep,
                                                      news_heads=title_heads, embed_dropout=False)
    valid = batch["candidate_mask"] == 1
    err = float(np.abs(scores - o_scores)[valid].max())
    print("v1 fp16 train %s p=%.1f: max |score - oracle| = %.3e, |loss diff| %.2e" % (case, p_drop, err, abs(loss - o_loss)))
    assert err < v1_bar(o_scores[valid])
    back = {v: k for k, v in zip(params.keys(), v0.keys())}
    _v1_grad_report(grads, o_grads, back, case)
    assert not grads[back["news_encoder.word_embedding.0.weight"]][0].any()


def test_v1_fp16_is_bit_reproducible_with_and_without_helper_streams(monkeypatch):
    """As tests/test_hip_v1.py::test_helper_streams_do_not_change_a_bit, for the fused fp16 kernels of the v1 news encoder:
    d(W_O) | d(b_o), d(W_add) (helper stream 0) and d(W_qkv) (helper stream 1) against a single-stream run, and two runs of the
    same configuration -- every gradient is a fixed-order sum (the all-padding titles' closed form included)."""
    from tests.test_hip_v1 import fwd_bwd
    shape = synth.Shape(n_words=4000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=24, history_len=50, n_candidates=5, n_words_title=20)
    batch = synth.make_batch(shape, seed=44, ragged=True, min_title=1, all_pad_title=True, mask_some_candidates=True)
    res = []
    for no_side in (False, True, False):
        if no_side:
            monkeypatch.setenv("NRMS_NO_SIDE_STREAMS", "1")
        else:
            monkeypatch.delenv("NRMS_NO_SIDE_STREAMS", raising=False)
        model = make_v1(shape, synth.make_params_v1(shape, seed=43), 6, precision="fp16").train()
        assert _news_precision(model, 8, 20, True) == _lib.NRMS_PRECISION_FP16
        res.append(fwd_bwd(model, batch))
        torch.cuda.synchronize()
    for other in (res[1], res[2]):
        assert np.array_equal(res[0][0], other[0])
        for n in res[0][2]:
            assert np.array_equal(res[0][2][n], other[2][n]), n


def test_v1_fp16_routes_what_the_fused_kernels_do_not_cover_to_bf16x3():
    """Masked encoders (nrms_v1.py:15-23,87-105: pair mask / additive mask -- the model's own forward passes none), titles longer
    than 32 words, heads of 32 columns or fewer with W_O: precision "fp16" runs them on the split-bf16 kernels (the masked
    primitives are tested against the oracle in tests/test_hip_v1.py in that mode); the C ABI refuses them in fp16."""
    import ctypes as C
    shape, title_heads, _ = V1_SHAPES["reference"]
    model = make_v1(shape, synth.make_params_v1(shape, seed=1), title_heads, precision="fp16").train()
    eng = model.engine
    BF = _lib.PRECISIONS["bf16x3"]
    assert eng._desc("news_encoder", 8, 20, mask_mode=1, training=True).precision == BF
    assert eng._desc("news_encoder", 8, 20, mask_mode=2).precision == BF
    assert eng._desc("news_encoder", 8, 40).precision == BF
    assert eng._desc("user_encoder", 8, 50).precision == BF
    assert eng._desc("news_encoder", 8, 20).precision == _lib.NRMS_PRECISION_FP16
    lib = _lib.load()
    for kw in (dict(mask_mode=1), dict(seq_len=40), dict(n_heads=10), dict(flags=0), dict(p_drop_embed=0.1)):
        args = dict(n_seq=4, seq_len=20, d_model=300, n_heads=6, q_dim=200, vocab=100, p_drop_embed=0.0, p_drop_ctx=0.0,
                    precision=_lib.NRMS_PRECISION_FP16, use_output_proj=1, mask_mode=0, flags=_lib.NRMS_FLAG_PAD_ROW_ZERO,
                    seed=0, loss_scale=0.0, p_drop_attn=0.0)
        args.update(kw)
        desc = _lib.EncoderDesc(**args)
        assert lib.nrms_encoder_fwd_scratch_bytes(C.byref(desc)) == 0, kw
        assert b"fp16" in lib.nrms_last_error(), kw


def test_v1_fp16_at_the_benchmarked_size():
    """bench.py's nrms_v1 leg (512 users x (50 + 5) titles of 20 words, 300 columns, six title heads of 50, ten user heads): the
    fused fp16 news encoder against the library's exact fp32 mode (itself pinned to the oracle at the oracle's sizes) -- every
    workgroup walks many title groups here (7 042 groups on 512 workgroups), the per-wave d(attn) scratch is reused, the title
    lists hold all three classes.  Scores inside north_star's 1e-4 on every valid score, every gradient tensor within 4e-3 of
    its scale, run-to-run determinism, user-permutation equivariance (bit-exact), dropout path finite."""
    from tests.test_hip_fullsize import noise_only
    shape = synth.Shape(n_words=synth.BENCH.n_words, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=512, history_len=50, n_candidates=5, n_words_title=20)
    params = synth.make_params_v1(shape, seed=0)
    batch = synth.make_batch(shape, seed=1, mask_some_candidates=True)
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    model = make_v1(shape, params, 6, precision="fp32")
    eng, flat, lay = model.engine, model._flat, model._layout

    def scores(training=True):
        return model.engine.forward(flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=training)

    ref = scores().clone()
    valid = tb["candidate_mask"] == 1
    _, dce = eng.ce_loss(ref, grad_scale=1.0 / shape.batch_size)
    g32 = torch.zeros_like(flat)
    eng.backward(flat, g32, dce)
    model.config.precision = "fp16"
    eng = model.engine
    assert eng.precision == "fp16" and _news_precision(model, 8, 20, True) == _lib.NRMS_PRECISION_FP16
    s1 = scores().clone()
    assert torch.equal(s1, scores())
    e = (s1 - ref)[valid].abs().double()
    srms, smax = float((ref[valid].double() ** 2).mean().sqrt()), float(ref[valid].abs().max())
    rms, emax = float((e * e).mean().sqrt()), float(e.max())
    print("v1 full size fp16 vs fp32 over %d scores of rms %.3f (max %.3f): rms %.2e, max %.2e, %.2f %% of the scores above 1e-4" % (
        e.numel(), srms, smax, rms, emax, 100.0 * float((e > 1e-4).double().mean())))
    # The v1 value path has seven fp16 roundings where v0's has four (x, W_V, V, P + the head concatenation, W_O, the projection's
    # output) on scores 1.5 x larger at this initialisation (rms 0.09, max 0.36): about 2 % of a 512-user batch's scores end up
    # further than north_star's ABSOLUTE 1e-4 from fp32 (max 1.56e-4) -- which is why this kernel family is OPT-IN for nrms_v1
    # (config.fp16_v1_news_encoder; the default keeps v1's news encoder in bf16x3, test below) and held here to its own stated
    # bound: 2e-4 absolute, relative rms below 6e-4
    assert emax < V1_FP16_OPT_IN_BAR and 1e-7 < rms < 6e-4 * srms
    g16 = torch.zeros_like(flat)
    scores()
    eng.backward(flat, g16, dce)
    g16b = torch.zeros_like(flat)
    scores()
    eng.backward(flat, g16b, dce)
    assert torch.equal(g16, g16b)
    for name in lay.names:
        a, b = lay.view(g16, name).double(), lay.view(g32, name).double()
        scale = float(b.abs().max())
        rel = float((a - b).abs().max()) / (scale + 1e-300)
        print("   v1 fp16 vs fp32 grad %-66s scale %.2e  max err %.1e of scale" % (name, scale, rel))
        v0name = name.replace("multi_head_self_attention.linear_layers.0", "multihead_self_attention.W_Q").replace(
            "multi_head_self_attention.linear_layers.1", "multihead_self_attention.W_K").replace(
            "multi_head_self_attention.linear_layers.2", "multihead_self_attention.W_V").replace(
            "additive_attention.query_vector", "additive_attention.attention_query_vector")
        if noise_only(v0name):
            continue
        assert rel < 4e-3, (name, rel)
    # a user's scores do not depend on its batch position
    perm = torch.from_numpy(np.random.default_rng(3).permutation(shape.batch_size)).cuda()
    tb_p = {k: v[perm] for k, v in tb.items()}
    s_perm = model.engine.forward(flat, tb_p["browsed_titles"], tb_p["candidate_titles"], tb_p["candidate_mask"], training=True)
    assert torch.equal(s_perm, s1[perm])
    sd = model.engine.forward(flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=True, p_drop=0.2, seed=11)
    assert torch.isfinite(sd[valid]).all()


def test_v1_fp16_deferred_wqkv_backward_equals_plain_backward():
    """The data-parallel split of the backward (NRMS_FLAG_DEFER_WQKV + nrms_encoder_bwd_wqkv: the table gradient is complete
    when the hook runs, the weight-gradient GEMMs of both helper streams -- d(W_add), d(W_O) | d(b_o) with the all-padding
    titles' closed form, d(W_qkv) -- are joined afterwards) equals the one-call backward bit for bit, v1 news encoder in fp16."""
    shape = synth.Shape(n_words=4000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=9, history_len=12, n_candidates=4, n_words_title=20)
    params = synth.make_params_v1(shape, seed=141)
    batch = synth.make_batch(shape, seed=142, ragged=True, min_title=1, all_pad_title=True)
    model = make_v1(shape, params, 6, precision="fp16").train()
    eng, flat, lay = model.engine, model._flat, model._layout
    assert _news_precision(model, 8, 20, True) == _lib.NRMS_PRECISION_FP16
    bt, ct, cm = (torch.from_numpy(batch[k]).cuda() for k in ("browsed_titles", "candidate_titles", "candidate_mask"))
    s = eng.forward(flat, bt, ct, cm, training=True)
    dsc = (torch.randn(s.shape, generator=torch.Generator().manual_seed(3)) * 1e-2).cuda()
    g_plain = torch.zeros_like(flat)
    eng.backward(flat, g_plain, dsc)
    eng.forward(flat, bt, ct, cm, training=True)
    g_split = torch.zeros_like(flat)
    seen = []

    def hook():
        torch.cuda.synchronize()
        seen.append(float(lay.view(g_split, "news_encoder.word_embedding.weight").abs().max()))

    eng.backward(flat, g_split, dsc, table_grad_ready=hook)
    assert len(seen) == 1 and seen[0] > 0.0
    for n in lay.names:
        assert torch.equal(lay.view(g_split, n), lay.view(g_plain, n)), n


def test_run_v0_entry_point_with_nrms_v1_in_fp16(tmp_path, monkeypatch):
    """The reference's entry contract (run_v0.py --model nrms_v1 -> model.Model(config, args) -> model.nrms_v1) with
    --precision fp16 on the synthetic corpus: MyDataset batches, autograd + torch.optim.Adam through the fused fp16 news encoder
    (the default title geometry of config.py: 20-word titles, six title heads), a dev evaluation, and -- if one was written --
    a checkpoint with the reference's v1 names behind the wrapper's prefix."""
    import os
    from pytorch_news_recommender_amd import run_v0
    monkeypatch.chdir(tmp_path)
    hist = run_v0.main(["--model", "nrms_v1", "--dataset", "synthetic", "--precision", "fp16", "--epochs", "1", "--synthetic_users", "256",
                        "--batch_size", "32", "--max_batches", "6", "--num_workers", "0", "--description", "T",
                        "--data_path", str(tmp_path / "data_processed"), "--save_path", str(tmp_path / "save")])
    assert len(hist["losses"]) == 6 and np.isfinite(hist["losses"]).all()
    assert hist["aucs"] and 0.0 < hist["aucs"][-1][1] < 1.0
    save = tmp_path / "save"
    for f in (os.listdir(save) if os.path.isdir(save) else []):       # (a checkpoint is written only when the dev AUC improves on 0.5)
        if f.endswith(".ckpt"):
            sd = torch.load(os.path.join(save, f), map_location="cpu", weights_only=True)
            assert "model.news_encoder.multi_head_self_attention.output_linear.weight" in sd


@pytest.mark.parametrize("variant", ["v0", "v1"])
def test_fp16_backward_without_the_kept_forward_scratch_rebuilds_the_same_lists(variant, monkeypatch):
    """NRMS_FLAG_FWD_SCRATCH_KEPT is an optimisation of the host driver (the backward reads the token / title lists its forward
    left in acts.scratch); a C-ABI caller that does not set it gets the lists rebuilt in the backward's workspace (and the
    context-dropout mask regenerated) -- the same gradients, bit for bit.  Both fused fp16 news encoders, dropout on."""
    from pytorch_news_recommender_amd import _lib as lib_mod
    from tests.test_hip_parity import make_model
    if variant == "v1":
        shape, title_heads, _ = V1_SHAPES["long_titles"]
        model = make_v1(shape, synth.make_params_v1(shape, seed=3), title_heads, dropout=0.2, precision="fp16").train()
    else:
        shape = synth.Shape(n_words=3000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                            batch_size=8, history_len=20, n_candidates=5, n_words_title=30)
        model = make_model(shape, synth.make_params(shape, seed=3), dropout=0.2, precision="fp16").train()
    batch = synth.make_batch(shape, seed=4, ragged=True, min_title=1, all_pad_title=True)
    eng, flat = model.engine, model._flat
    bt, ct, cm = (torch.from_numpy(batch[k]).cuda() for k in ("browsed_titles", "candidate_titles", "candidate_mask"))
    grads = []
    for kept in (True, False):
        if not kept:
            monkeypatch.setattr(lib_mod, "NRMS_FLAG_FWD_SCRATCH_KEPT", 0)
        s = eng.forward(flat, bt, ct, cm, training=True, p_drop=0.2, seed=77)
        dsc = (torch.randn(s.shape, generator=torch.Generator().manual_seed(5)) * 1e-2).cuda()
        g = torch.zeros_like(flat)
        eng.backward(flat, g, dsc)
        grads.append(g)
    assert float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0], grads[1])


def test_a_stale_kept_scratch_promise_is_noticed_on_the_host():
    """with NRMS_FLAG_FWD_SCRATCH_KEPT the backward took the token / title lists from acts.scratch unchecked; a caller
    that had handed the same scratch to ANOTHER forward in between (other ids, another n_seq) got lists that index out of bounds.
    The library now records on the host which forward built the lists of a scratch buffer, forgets it when any forward is given
    that buffer again, and rebuilds the lists when the record does not match the backward's arguments: the gradients of a
    training forward whose scratch was reused equal those of an undisturbed one, bit for bit."""
    from tests.test_hip_parity import make_model
    shape = synth.Shape(n_words=3000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=8, history_len=20, n_candidates=5, n_words_title=30)
    model = make_model(shape, synth.make_params(shape, seed=3), dropout=0.2, precision="fp16").train()
    batch = synth.make_batch(shape, seed=4, ragged=True, min_title=1, all_pad_title=True)
    eng, flat = model.engine, model._flat
    bt, ct, cm = (torch.from_numpy(batch[k]).cuda() for k in ("browsed_titles", "candidate_titles", "candidate_mask"))
    other = torch.from_numpy(synth.make_batch(shape, seed=9, ragged=True, min_title=1)["browsed_titles"]).cuda().reshape(-1, 30)[:37].contiguous()
    grads = []
    for disturb in (False, True):
        s = eng.forward(flat, bt, ct, cm, training=True, p_drop=0.2, seed=77)
        if disturb:
            # another news-encoder forward on the SAME scratch memory (its own activation buffers: only the lists are clobbered)
            eng._bufs["intruder.scratch16"] = eng._bufs["news.scratch16"]
            eng.encode_titles(flat, other, save=True, tag="intruder", p_ctx=0.2, seed=5)
        dsc = (torch.randn(s.shape, generator=torch.Generator().manual_seed(5)) * 1e-2).cuda()
        g = torch.zeros_like(flat)
        eng.backward(flat, g, dsc)
        torch.cuda.synchronize()
        grads.append(g)
    assert float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0], grads[1])


def test_v1_default_routing_against_the_oracle_at_the_benchmarked_size():
    """nrms_v1 with config.precision = "fp16" and nothing else set -- what bench.py's nrms_v1 leg leads with: the news encoder
    (six heads of 50, W_O) stays on the split-bf16 kernels because its fused fp16 form misses the absolute 1e-4 (above), so the
    whole model is bf16x3.  Scores of the 512-user bench batch against the ORACLE's (one batched CPU forward): 2e-5 in the
    training forward and in inference; the opt-in fp16 news encoder is measured beside it against the same oracle scores."""
    import time
    from oracle import nrms_oracle as orc
    shape = synth.Shape(n_words=synth.BENCH.n_words, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=512, history_len=50, n_candidates=5, n_words_title=20)
    params = synth.make_params_v1(shape, seed=0)
    batch = synth.make_batch(shape, seed=1, mask_some_candidates=True)
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    t0 = time.time()
    with torch.no_grad():
        o_scores, _ = orc.forward(orc.to_torch(orc.v1_to_v0_names(params)), batch, shape.num_attention_heads, news_heads=6,
                                  embed_dropout=False)
    o_scores = o_scores.numpy()
    print("oracle v1 forward at 512 users: %.1f s" % (time.time() - t0))
    valid = batch["candidate_mask"] == 1
    model = make_v1(shape, params, 6, precision="fp16", fp16_news=False, fp16_inference=False)
    BF = _lib.PRECISIONS["bf16x3"]
    assert _news_precision(model, 8, 20, True) == BF and _news_precision(model, 8, 20, False) == BF
    flat = model._flat
    for training in (True, False):
        s = model.engine.forward(flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=training).cpu().numpy()
        err = float(np.abs(s - o_scores)[valid].max())
        print("v1 default routing (bf16x3) vs ORACLE, training=%s: max |score diff| %.2e over %d scores (max |score| %.3f)" % (
            training, err, int(valid.sum()), float(np.abs(o_scores[valid]).max())))
        assert err < 2e-5
    model.config.fp16_v1_news_encoder = True
    assert _news_precision(model, 8, 20, True) == _lib.NRMS_PRECISION_FP16
    s = model.engine.forward(flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=True).cpu().numpy()
    e = np.abs(s - o_scores)[valid]
    print("v1 OPT-IN fp16 news encoder vs ORACLE: rms %.2e max %.2e, %.2f %% of the scores beyond 1e-4" % (
        float(np.sqrt((e.astype(np.float64) ** 2).mean())), float(e.max()), 100.0 * float((e > 1e-4).mean())))
    assert float(e.max()) < V1_FP16_OPT_IN_BAR
