"""nrms_naml on the GPU, through the drop-in Model and the C ABI: fixture g7 (outputs of the
imported reference's ``model.nrms_naml.Model``: odd widths and the real 300 / 800 widths), the oracle on seeded inputs,
a dropout replay (the keep masks the kernels used -- attention probabilities and feature rows -- fed to the oracle), LayerNorm
on its own and the fused train step.

Tolerances: scores 2e-5 (fp32) / 1e-4 (bf16x3, north_star's bar; measured 5e-5 on scores of magnitude 9) absolute; gradients |got - ref| <= rtol |ref| + atol + scale * max|ref| per tensor
(fp32: summation order; bf16x3: ~2^-16 relative per product)."""
import os

import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import synth

pytestmark = pytest.mark.gpu

MODES = ["fp32", "bf16x3"]
TOL = {"fp32": dict(score=2e-5, rtol=1e-3, atol=2e-6, scale=3e-6),
       "bf16x3": dict(score=1e-4, rtol=1e-3, atol=2e-6, scale=4e-5)}     # scores here are O(10): 1e-4 = 1e-5 relative


def make_model(shape, params, dropout=0.0, precision="fp32"):
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.model.nrms_naml_hip import Model
    cfg = Config("nrms_naml")
    cfg.__nrms__()
    for k in ("word_embed_size", "title_heads_num", "query_vector_dim", "category_nums", "subcategory_nums",
              "cate_embed_size", "user_heads_num", "query_vector_dim_large"):
        setattr(cfg, k, getattr(shape, k))
    cfg.news_feature_size = shape.news_feature_size
    cfg.dropout = dropout
    cfg.precision = precision
    m = Model(cfg, pretrained_word_embedding=params["news_encoder.word_embedding.weight"])
    res = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    assert not res.missing_keys and not res.unexpected_keys
    return m.to("cuda")


def tbatch(batch):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}


def fwd_bwd(model, batch):
    model.zero_grad()
    scores = model(tbatch(batch))
    loss = torch.nn.CrossEntropyLoss()(scores, torch.zeros(len(scores), dtype=torch.long, device=scores.device))
    loss.backward()
    grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}
    return scores.detach().cpu().numpy(), float(loss.detach()), grads


def close(got, ref, t, name):
    ref = np.asarray(ref)
    bound = t["rtol"] * np.abs(ref) + t["atol"] + t["scale"] * float(np.abs(ref).max() if ref.size else 0.0)
    diff = np.abs(np.asarray(got) - ref)
    worst = float((diff - bound).max()) if diff.size else 0.0
    assert worst <= 0.0, "%s: max |diff| %.3e exceeds the bound by %.3e (scale %.3e)" % (
        name, float(diff.max()), worst, float(np.abs(ref).max()))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tag", ["odd", "mind"])
def test_g7_forward_backward(golden_dir, tag, mode):
    from tests.test_oracle_golden import naml_sample_rows
    g = np.load(os.path.join(golden_dir, "g7_naml.npz"))
    shape = synth.G7_ODD if tag == "odd" else synth.G7_MIND
    params = synth.make_params_naml(shape, seed=21)
    batch = synth.make_batch_naml(shape, seed=22)
    model = make_model(shape, params, precision=mode).train()            # dropout 0, train mode: as the fixture
    assert list(model.state_dict().keys()) == list(params.keys())        # the reference's names in the reference's order
    t = TOL[mode]
    scores, loss, grads = fwd_bwd(model, batch)
    live = batch["candidate_mask"] != 0
    np.testing.assert_allclose(scores[live], g[tag + "/scores"][live], rtol=0, atol=t["score"])
    assert (scores[~live] == np.float32(-1e9)).all()
    assert abs(loss - float(g[tag + "/loss"])) < t["score"]
    for n in params:
        if tag + "/grad/" + n in g:
            close(grads[n], g[tag + "/grad/" + n], t, n)
        else:
            close(grads[n][naml_sample_rows(grads[n].shape[0])], g[tag + "/grad_rows/" + n], t, n)
            loose = dict(t, atol=t["atol"] * 30, scale=t["scale"] * 30)
            close(grads[n].sum(1, dtype=np.float64), g[tag + "/grad_rowsum/" + n], loose, n + " row sums")
            close(grads[n].sum(0, dtype=np.float64), g[tag + "/grad_colsum/" + n], loose, n + " column sums")
    for n in ("news_encoder.word_embedding.weight", "news_encoder.category_embedding.weight",
              "news_encoder.subcategory_embedding.weight"):
        assert not grads[n][0].any(), n
    # eval mode = the same numbers (dropout is 0), through the inference path
    model.eval()
    with torch.no_grad():
        s_eval = model(tbatch(batch)).cpu().numpy()
    np.testing.assert_allclose(s_eval[live], g[tag + "/scores"][live], rtol=0, atol=t["score"])


def naml_keep_masks(model, shape, batch, seed, p):
    """The keep masks of one training forward, in the oracle's layout (sites 2 and 3 of nrms_dropout_keep_mask)."""
    eng = model.engine
    B, H, C = batch["browsed_titles"].shape[0], shape.history_len, shape.n_candidates
    N = B * (H + C)

    def flat_mask(s, site, total):
        k = eng.dropout_keep_mask(s & 0xFFFFFFFFFFFFFFFF, site, (total + 3) // 4, p, d=4)
        return k.reshape(-1)[:total]

    def split(m):                                   # slots: history rows first, then candidates
        return {"hist": m[:B * H], "cand": m[B * H:]}

    h = shape.title_heads_num
    Lt, La, F = shape.n_words_title, shape.n_words_abst, shape.news_feature_size
    ta = split(flat_mask(seed, 2, N * h * Lt * Lt).view(N, h, Lt, Lt))
    aa = split(flat_mask(seed ^ 0x5DEECE66D1CE4E5B, 2, N * h * La * La).view(N, h, La, La))
    ft = split(flat_mask(seed, 3, N * F).view(N, F))
    ua = flat_mask(seed ^ 0x2545F4914F6CDD1D, 2, B * shape.user_heads_num * H * H).view(B, shape.user_heads_num, H, H)
    keep = {k: {"title_attn": ta[k].cpu(), "abst_attn": aa[k].cpu(), "feat": ft[k].cpu()} for k in ("hist", "cand")}
    keep["user_attn"] = ua.cpu()
    return keep


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("geom", [(96, 17), (204, 37)], ids=["dk16_S17", "dk34_S37"])
def test_dropout_replay_against_oracle(mode, geom):
    """Training forward + backward with dropout 0.2: the masks the kernels drew, replayed through the oracle.  Second
    geometry: 37-word abstracts with 34-wide heads -- the two-wave 64 x 64 attention units with the probability-dropout
    mask of a sequence length that is not a multiple of 4 (a lane's four keys straddle two Philox groups)."""
    from oracle import naml_oracle as nml
    d_word, n_abst = geom
    shape = synth.NamlShape(n_words=120, word_embed_size=d_word, title_heads_num=6, query_vector_dim=40, category_nums=7,
                            subcategory_nums=11, cate_embed_size=32, user_heads_num=8, query_vector_dim_large=72,
                            batch_size=4, history_len=9, n_candidates=4, n_words_title=10, n_words_abst=n_abst)
    params = synth.make_params_naml(shape, seed=5)
    batch = synth.make_batch_naml(shape, seed=6)
    p = 0.2
    model = make_model(shape, params, dropout=p, precision=mode).train()
    scores, loss, grads = fwd_bwd(model, batch)
    seed = model.engine._saved["seed"]
    keep = naml_keep_masks(model, shape, batch, seed, p)
    frac = float(keep["hist"]["feat"].float().mean())
    assert 0.7 < frac < 0.9
    ref_scores, ref_loss, ref_grads = nml.loss_and_grads(params, batch, shape.title_heads_num, shape.user_heads_num,
                                                         p_drop=p, keep=keep)
    t = TOL[mode]
    live = batch["candidate_mask"] != 0
    np.testing.assert_allclose(scores[live], ref_scores[live], rtol=0, atol=t["score"])
    assert abs(loss - ref_loss) < t["score"]
    for n in params:
        close(grads[n], ref_grads[n], t, n)
    # a second forward draws other masks
    s2 = model(tbatch(batch)).detach().cpu().numpy()
    assert np.abs(s2[live] - scores[live]).max() > 1e-3


def test_layernorm_alone():
    """nrms_layernorm_fwd / _bwd against torch on the GPU tensors (the feature rows are covered by g7 and the replay)."""
    import ctypes as C
    from pytorch_news_recommender_amd import _lib
    shape = synth.G7_ODD
    model = make_model(shape, synth.make_params_naml(shape, seed=3))
    eng, flat = model.engine, model._flat
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    F = shape.news_feature_size
    x = (torch.randn(37, F, generator=g) * 3 + 1).to(dev)
    stats = torch.empty(37, 2, device=dev)
    y = eng.layernorm(flat, x, stats)
    w, b = model.norm.weight.detach(), model.norm.bias.detach()
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (F,), wr, br, 1e-5)
    np.testing.assert_allclose(y.cpu().numpy(), yr.detach().cpu().numpy(), rtol=0, atol=3e-6)
    dy = torch.randn(37, F, generator=g).to(dev)
    yr.backward(dy)
    dx = torch.empty_like(x)
    dgb = torch.zeros(2 * F, device=dev)
    ws = torch.empty(int(eng.lib.nrms_layernorm_bwd_workspace_bytes(F)) // 4, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = eng.lib.nrms_layernorm_bwd(C.c_int64(37), F, _lib.ptr(x), _lib.ptr(w.contiguous()), _lib.ptr(stats), _lib.ptr(dy),
                                    _lib.ptr(dx), _lib.ptr(dgb), _lib.ptr(ws), C.c_size_t(ws.numel() * 4), stream)
    _lib.check(rc, "nrms_layernorm_bwd")
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(dgb[:F].cpu().numpy(), wr.grad.cpu().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dgb[F:].cpu().numpy(), br.grad.cpu().numpy(), rtol=1e-4, atol=2e-5)
    # workspace too small -> NRMS_EWORKSPACE, reported as NrmsError
    rc = eng.lib.nrms_layernorm_bwd(C.c_int64(37), F, _lib.ptr(x), _lib.ptr(w.contiguous()), _lib.ptr(stats), _lib.ptr(dy),
                                    _lib.ptr(dx), _lib.ptr(dgb), _lib.ptr(ws), C.c_size_t(16), stream)
    with pytest.raises(_lib.NrmsError):
        _lib.check(rc, "nrms_layernorm_bwd")


@pytest.mark.parametrize("mode", MODES)
def test_fused_train_step_is_forward_backward_adam(mode):
    """Model.train_step (forward, CE, backward, Adam on the flat buffers) against the same model's autograd gradients
    (pinned to the reference by the tests above) pushed through oracle.adam_step: the first Adam step turns a gradient
    into +-lr whatever its size, so the comparison uses identical gradients (Adam itself: test_adam_step_kernel_alone)."""
    from oracle import naml_oracle as nml
    from oracle import nrms_oracle as orc
    shape = synth.G7_ODD
    params = synth.make_params_naml(shape, seed=31)
    batch = synth.make_batch_naml(shape, seed=32)
    _, loss, grads = fwd_bwd(make_model(shape, params, precision=mode).train(), batch)
    model = make_model(shape, params, precision=mode).train()
    loss_sum = model.train_step(tbatch(batch), lr=1e-3)
    _, ref_loss, _ = nml.loss_and_grads(params, batch, shape.title_heads_num, shape.user_heads_num)
    assert abs(float(loss_sum) / shape.batch_size - ref_loss) < TOL[mode]["score"]
    assert abs(float(loss_sum) / shape.batch_size - loss) < 1e-6
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for n, p0 in params.items():
        want = p0.astype(np.float64)
        orc.adam_step(want, grads[n].astype(np.float64), np.zeros_like(want), np.zeros_like(want), 1)     # in place
        if n.endswith("linear_layers.1.bias"):
            continue        # d(b_K) is identically zero in exact arithmetic (softmax is shift invariant): rounding noise
        solid = np.abs(grads[n]) > 1e-6          # below that Adam's eps = 1e-8 makes the update sensitive to the last bits of g
        np.testing.assert_allclose(sd[n][solid], want[solid], rtol=0, atol=3e-7, err_msg=n)
        assert np.abs(sd[n] - p0).max() <= 1.001e-3
    # a second step runs (optimizer state carried) and keeps the padding rows where they were
    model.train_step(tbatch(batch), lr=1e-3)
    for n in ("news_encoder.word_embedding.weight", "news_encoder.category_embedding.weight",
              "news_encoder.subcategory_embedding.weight"):
        assert torch.equal(model.state_dict()[n][0].cpu(), torch.from_numpy(params[n][0])), n


def test_run_v0_entry_point_with_nrms_naml(tmp_path, monkeypatch):
    """The reference's entry contract (run_v0.py --model nrms_naml -> model.Model(config, args) -> model.nrms_naml) on the
    synthetic corpus: data_handler.MyDataset feeds titles, abstracts and category ids; a few training steps, a dev
    evaluation, a checkpoint whose keys are the reference's behind the wrapper's ``model.`` prefix."""
    from pytorch_news_recommender_amd import run_v0
    monkeypatch.chdir(tmp_path)
    hist = run_v0.main(["--model", "nrms_naml", "--dataset", "synthetic", "--epochs", "1", "--synthetic_users", "192",
                        "--batch_size", "32", "--max_batches", "5", "--num_workers", "0", "--description", "T",
                        "--data_path", str(tmp_path / "data_processed"), "--save_path", str(tmp_path / "save")])
    assert len(hist["losses"]) == 5 and np.isfinite(hist["losses"]).all()
    assert hist["aucs"] and 0.0 < hist["aucs"][-1][1] < 1.0
    ckpts = [f for f in os.listdir(tmp_path / "save") if f.endswith(".ckpt")]
    assert ckpts
    sd = torch.load(os.path.join(tmp_path / "save", ckpts[0]), map_location="cpu", weights_only=True)
    assert "model.norm.weight" in sd and "model.news_encoder.category_embedding.weight" in sd
    assert sd["model.user_encoder.additive_attention.linear.weight"].shape == (400, 800)


def test_inference_over_distinct_news_equals_every_slot():
    """Evaluation-shaped batch (many repeated news items and padding slots): the grouped path (one feature row per distinct
    (title, abstract, category, sub-category)) against encoding every slot, and against the oracle."""
    from oracle import naml_oracle as nml
    shape = synth.NamlShape(n_words=90, word_embed_size=48, title_heads_num=6, query_vector_dim=20, category_nums=5,
                            subcategory_nums=9, cate_embed_size=16, user_heads_num=8, query_vector_dim_large=36,
                            batch_size=6, history_len=12, n_candidates=40, n_words_title=7, n_words_abst=11)
    params = synth.make_params_naml(shape, seed=41)
    batch = synth.make_batch_naml(shape, seed=42)
    # impressions padded to 40 slots: 8 shown, the rest padding; histories repeat two news items
    for k in ("candidate_titles", "candidate_absts"):
        batch[k][:, 8:, :] = 0
    for k in ("candidate_categ_ids", "candidate_subcateg_ids"):
        batch[k][:, 8:] = 0
    batch["candidate_mask"][:, 8:] = 0
    for k in ("browsed_titles", "browsed_absts", "browsed_categ_ids", "browsed_subcateg_ids"):
        batch[k][:, 2:] = batch[k][:, :1].repeat(shape.history_len - 2, axis=1)
    model = make_model(shape, params).eval()
    with torch.no_grad():
        model.dedup_inference = True
        s_dedup = model(tbatch(batch)).cpu().numpy()
        n_unique = model.engine.last_unique_news
        model.dedup_inference = False
        s_plain = model(tbatch(batch)).cpu().numpy()
    assert n_unique < 0.3 * shape.batch_size * (shape.history_len + shape.n_candidates)
    live = batch["candidate_mask"] != 0
    np.testing.assert_allclose(s_dedup[live], s_plain[live], rtol=0, atol=2e-6)
    assert (s_dedup[~live] == np.float32(-1e9)).all()
    p = nml.to_torch(params)
    with torch.no_grad():
        ref = nml.forward(p, tbatch(batch), shape.title_heads_num, shape.user_heads_num).numpy()
    np.testing.assert_allclose(s_dedup[live], ref[live], rtol=0, atol=2e-5)


@pytest.mark.parametrize("mode", MODES)
def test_padding_token_skip_equals_dense_path_with_dropout(mode):
    """nrms_naml at the reference's widths (d = 300, 6 heads of 50, 20-word titles, 40-word abstracts: the two-wave 64 x 64
    attention units) with dropout 0.2: the padding-skipping path -- compact Q|K|V / d(W_qkv) / dX rows, all-padding
    sequences through the kept-key closed form -- against the dense path (skip_padding_tokens = False) drawing the same
    masks.  Same function, different summation orders: scores and every gradient agree to rounding."""
    shape = synth.NamlShape(n_words=300, batch_size=4, history_len=12, n_candidates=3)
    assert shape.word_embed_size == 300 and shape.n_words_abst == 40 and shape.title_heads_num == 6
    params = synth.make_params_naml(shape, seed=15)
    batch = synth.make_batch_naml(shape, seed=16)
    n_empty = int((batch["browsed_absts"].reshape(-1, shape.n_words_abst) == 0).all(1).sum())
    assert n_empty >= 5                                   # empty history slots: the closed form is exercised
    out = []
    for skip in (True, False):
        torch.manual_seed(1234)
        model = make_model(shape, params, dropout=0.2, precision=mode).train()
        model.config.skip_padding_tokens = skip
        res = fwd_bwd(model, batch)
        assert model.engine.pad_row_zero == skip
        out.append(res)
    (s1, l1, g1), (s0, l0, g0) = out
    live = batch["candidate_mask"] != 0
    tol = 2e-5 if mode == "fp32" else 1e-4
    assert np.abs(s1[live] - s0[live]).max() < tol, np.abs(s1[live] - s0[live]).max()
    assert abs(l1 - l0) < tol
    for n in g0:
        if n.endswith("linear_layers.1.bias"):            # the K bias: an analytically zero gradient, rounding noise in both
            continue
        scale = float(np.abs(g0[n]).max())
        rel = 2e-5 if mode == "fp32" else 2e-4
        if n.endswith("additive_attention.linear.bias"):  # sum_s ds_s = 0: a cancelling sum, ~1e-2 relative noise
            rel = 2e-2
        if n.endswith("additive_attention.query_vector"):  # sum_s ds_s t_s with sum_s ds_s = 0 and nearly equal t rows in an all-padding
            rel = max(rel, 1e-4)                           # sequence: the closed form and the GEMM chain round it differently (2.5e-5)
        assert np.abs(g1[n] - g0[n]).max() <= rel * scale + 1e-9, (n, float(np.abs(g1[n] - g0[n]).max()), scale)


@pytest.mark.parametrize("geom", [(300, 6, 200, 20), (300, 6, 200, 40), (96, 6, 40, 17), (64, 2, 12, 1)], ids=["title20", "abst40", "dk16_S17", "one_word"])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_all_padding_closed_form_equals_the_chain(geom, p):
    """nrms_encoder_empty_fwd / _bwd (csrc/empty_seq.hip) against the kernel chain on a batch of all-padding sequences: the chain
    processes such sequences itself (attention through the kept-key closed form, then W_O and the additive attention row by row),
    so both must give the same vectors and the same gradient of every weight -- with probability dropout too, both drawing the
    decisions of the sequence numbers in seq_index (desc.seq_index for the chain: what a caller that compacts its batch passes)."""
    import ctypes as C
    from pytorch_news_recommender_amd import _lib
    d, h, q, S = geom
    shape = synth.NamlShape(n_words=50, word_embed_size=d, title_heads_num=h, query_vector_dim=q, category_nums=5, subcategory_nums=9,
                            cate_embed_size=16, user_heads_num=8, query_vector_dim_large=36, batch_size=2, history_len=3, n_candidates=2,
                            n_words_title=S, n_words_abst=S)
    params = synth.make_params_naml(shape, seed=21)
    model = make_model(shape, params, dropout=p, precision="fp32").train()
    eng, flat, lib = model.engine, model._flat, _lib.load()
    assert eng.pad_row_zero
    n = 37
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ids = torch.zeros(n, S, dtype=torch.int64, device="cuda")
    sidx = (torch.randperm(500, generator=torch.Generator().manual_seed(1))[:n]).to(torch.int32).cuda()    # their numbers in a "full batch"
    dout = (torch.randn(n, d, generator=torch.Generator().manual_seed(2)) * 0.1).cuda()
    seed = 0x1234567
    w = eng._ptrs(_lib.EncoderWeights, flat, "news_encoder")
    desc = eng._desc("news_encoder", n, S, p, seed)
    desc.seq_index = sidx.data_ptr()
    acts = eng._acts("t_empty", desc, gather=True)
    out_chain = torch.empty(n, d, device="cuda")
    _lib.check(lib.nrms_encoder_fwd(C.byref(desc), C.byref(w), _lib.ptr(ids), None, None, C.byref(acts), _lib.ptr(out_chain), stream), "fwd")
    g_chain = torch.zeros_like(flat)
    ws = eng._bwd_workspace(desc)
    _lib.check(lib.nrms_encoder_bwd(C.byref(desc), C.byref(w), _lib.ptr(ids), None, None, C.byref(acts), _lib.ptr(dout),
                                    C.byref(eng._ptrs(_lib.EncoderGrads, g_chain, "news_encoder")), None, _lib.ptr(ws), C.c_size_t(ws.numel() * 4),
                                    stream), "bwd")
    desc_e = eng._desc("news_encoder", n, S, p, seed)
    ews = eng._empty_ws(desc_e)
    out_e = torch.empty(n, d, device="cuda")
    saved = torch.empty(lib.nrms_encoder_empty_saved_bytes(C.byref(desc_e)) // 4, device="cuda")
    _lib.check(lib.nrms_encoder_empty_fwd(C.byref(desc_e), C.byref(w), _lib.ptr(sidx), _lib.ptr(out_e), _lib.ptr(saved), _lib.ptr(ews),
                                          C.c_size_t(ews.numel() * 4), stream), "empty_fwd")
    out_inf = torch.empty(n, d, device="cuda")                 # inference form: nothing saved, same vectors
    _lib.check(lib.nrms_encoder_empty_fwd(C.byref(desc_e), C.byref(w), _lib.ptr(sidx), _lib.ptr(out_inf), None, _lib.ptr(ews),
                                          C.c_size_t(ews.numel() * 4), stream), "empty_fwd")
    assert torch.equal(out_inf, out_e)
    g_e = torch.zeros_like(flat)
    _lib.check(lib.nrms_encoder_empty_bwd(C.byref(desc_e), C.byref(w), _lib.ptr(sidx), _lib.ptr(dout), _lib.ptr(saved),
                                          C.byref(eng._ptrs(_lib.EncoderGrads, g_e, "news_encoder")), _lib.ptr(ews), C.c_size_t(ews.numel() * 4), stream), "empty_bwd")
    torch.cuda.synchronize()
    a, b = out_chain.cpu().numpy(), out_e.cpu().numpy()
    assert np.abs(a).max() > 1e-3
    assert np.abs(a - b).max() <= 2e-6 * max(1.0, float(np.abs(a).max())), np.abs(a - b).max()
    if p > 0:
        assert np.abs(a[0] - a[1]).max() > 1e-6            # different sequences drew different masks
    lay = model._layout
    for name in lay.names:
        if not name.startswith("news_encoder."):
            continue
        ga, gb = lay.view(g_chain, name).cpu().numpy(), lay.view(g_e, name).cpu().numpy()
        scale = float(np.abs(ga).max())
        # (the chain's d(b_q), d(b_k) of such sequences are rounding noise around an exact zero; the closed form leaves them at 0)
        tol = 3e-5 * scale + 1e-8 if scale > 1e-6 else 1e-6
        assert np.abs(ga - gb).max() <= tol, (name, float(np.abs(ga - gb).max()), scale)
    # second call accumulates
    _lib.check(lib.nrms_encoder_empty_bwd(C.byref(desc_e), C.byref(w), _lib.ptr(sidx), _lib.ptr(dout), _lib.ptr(saved),
                                          C.byref(eng._ptrs(_lib.EncoderGrads, g_e, "news_encoder")), _lib.ptr(ews), C.c_size_t(ews.numel() * 4), stream), "empty_bwd")
    n_wa = "news_encoder.additive_attention.linear.weight"
    assert np.allclose(lay.view(g_e, n_wa).cpu().numpy(), 2 * lay.view(g_chain, n_wa).cpu().numpy(), rtol=1e-4, atol=1e-7)


def test_sequence_partition_lists_and_seq_index_rules():
    """nrms_sequence_partition: the two lists (sequences with a real token / all-padding ones, ascending) and their sizes;
    desc.seq_index is refused where the counters it renumbers do not exist or would not be the only ones to move."""
    import ctypes as C
    from pytorch_news_recommender_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n, S = 1000, 20
    ids = rng.integers(1, 50, size=(n, S), dtype=np.int64)
    empty = rng.random(n) < 0.4
    ids[empty] = 0
    ids[~empty, 5:] = 0                                   # padding tails do not make a sequence all-padding
    d_ids = torch.from_numpy(ids).cuda()
    order = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    counts = torch.zeros(lib.nrms_sequence_partition_count_ints(n) + 2, dtype=torch.int32, device="cuda")
    _lib.check(lib.nrms_sequence_partition(_lib.ptr(d_ids), n, S, _lib.ptr(order), _lib.ptr(counts), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "partition")
    o, c = order.cpu().numpy(), counts.cpu().numpy()
    assert c[0] == int((~empty).sum()) and c[1] == int(empty.sum())
    assert np.array_equal(o[:c[0]], np.flatnonzero(~empty)) and np.array_equal(o[n:n + c[1]], np.flatnonzero(empty))
    base = dict(n_seq=4, seq_len=20, d_model=300, n_heads=6, q_dim=200, vocab=100, p_drop_embed=0.0, p_drop_ctx=0.0,
                precision=_lib.NRMS_PRECISION_BF16X3, use_output_proj=1, mask_mode=0, flags=_lib.NRMS_FLAG_PAD_ROW_ZERO, seed=1, loss_scale=0.0,
                p_drop_attn=0.2, seq_index=order.data_ptr())
    assert lib.nrms_encoder_bwd_workspace_bytes(C.byref(_lib.EncoderDesc(**base))) > 0
    for kw in (dict(p_drop_ctx=0.1), dict(p_drop_embed=0.1), dict(n_heads=3, flags=0), dict(precision=_lib.NRMS_PRECISION_FP16, use_output_proj=0, n_heads=10, p_drop_attn=0.0)):
        args = dict(base)
        args.update(kw)
        assert lib.nrms_encoder_bwd_workspace_bytes(C.byref(_lib.EncoderDesc(**args))) == 0, kw
        assert b"seq_index" in lib.nrms_last_error(), (kw, lib.nrms_last_error())


_FULL = {}


@pytest.mark.parametrize("mode", MODES)
def test_naml_against_the_oracle_at_the_benchmarked_size(mode):
    """512 users at the reference's own widths (title 20 / abstract 40 words, d = 300, 800-wide user encoder) -- the batch
    bench.py's nrms_naml leg times -- against the ORACLE's batched CPU forward: the training-mode path (dropout 0: the sequence
    partition, the kernel chain on the compacted 59 %, the closed form on 23 000 all-padding sequences) and the evaluation path
    (distinct news only).  Scores are O(10): the absolute 1e-4 of the bf16x3 row is 1e-5 relative."""
    import time
    from oracle import naml_oracle as nml
    shape = synth.NamlShape(batch_size=512)
    params = synth.make_params_naml(shape, seed=0)
    batch = synth.make_batch_naml(shape, seed=1)
    if "ref" not in _FULL:                              # one CPU forward (tens of seconds) for both modes
        t0 = time.time()
        with torch.no_grad():
            _FULL["ref"] = nml.forward(nml.to_torch(params), tbatch(batch), shape.title_heads_num, shape.user_heads_num).numpy()
        print("naml oracle forward at 512 users: %.1f s" % (time.time() - t0))
    ref = _FULL["ref"]
    live = batch["candidate_mask"] != 0
    model = make_model(shape, params, dropout=0.0, precision=mode)
    model.train()                                       # dropout 0: the training forward's kernels, deterministic
    with torch.no_grad():
        s_train = model(tbatch(batch)).cpu().numpy()
    eng = model.engine
    (order_t, n_t), (order_a, n_a) = eng.last_split
    N = 512 * 55
    assert 0.5 * N < n_t < 0.7 * N and 0.5 * N < n_a < 0.7 * N    # the all-padding history slots took the closed form
    model.eval()
    with torch.no_grad():
        s_eval = model(tbatch(batch)).cpu().numpy()
    for name, s in (("train-mode", s_train), ("eval", s_eval)):
        err = float(np.abs(s - ref)[live].max())
        print("naml %-6s %-10s max |score - oracle| = %.2e over %d scores (max |score| %.1f)" % (mode, name, err, int(live.sum()), float(np.abs(ref[live]).max())))
        assert err <= TOL[mode]["score"] * 1.5, (name, err)
