"""nrms_v1 semantics on the HIP path (SURVEY 8a rows a-3', a-4 mask variant, f-3): output projection
W_O, per-encoder head counts (6 news / 10 user), dropout only after the attention block, pairwise
attention mask and masked additive attention.  The oracle's v1 primitives are pinned to the
reference's own classes by fixture g3 (tests/test_oracle_golden.py::test_g3_v1_semantics)."""
import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import synth

from tests.test_hip_parity import MODES, TOL, assert_grad_close

pytestmark = pytest.mark.gpu


def make_v1(shape, params, title_heads, dropout=0.0, precision="fp32", fp16_news=True, fp16_inference=True):
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.model.nrms_v1_hip import Model
    cfg = Config("nrms_v1")
    cfg.__nrms__()
    cfg.word_embed_size, cfg.query_vector_dim = shape.word_embed_size, shape.query_vector_dim
    cfg.num_attention_heads, cfg.title_heads_num = shape.num_attention_heads, title_heads
    cfg.dropout = dropout
    cfg.precision = precision
    # precision "fp16" only: both switches are OPT-IN in the product (config.py); the kernel tests turn them on
    cfg.fp16_v1_news_encoder, cfg.fp16_inference = fp16_news, fp16_inference
    m = Model(cfg, pretrained_word_embedding=params["news_encoder.word_embedding.weight"])
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return m.to("cuda")


def fwd_bwd(model, batch):
    model.zero_grad()
    scores = model({k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()})
    loss = torch.nn.functional.cross_entropy(scores, torch.zeros(len(scores), dtype=torch.long, device=scores.device))
    loss.backward()
    return scores.detach().cpu().numpy(), float(loss.detach()), {n: p.grad.cpu().numpy() for n, p in model.named_parameters()}


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape,title_heads", [
    (synth.Shape(n_words=800, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                 batch_size=6, history_len=50, n_candidates=5, n_words_title=20), 6),       # res_logs.md:4 setup: L=20, h=6/10
    (synth.Shape(n_words=90, word_embed_size=48, num_attention_heads=4, query_vector_dim=16,
                 batch_size=3, history_len=5, n_candidates=2, n_words_title=7), 2),
    (synth.Shape(n_words=200, word_embed_size=100, num_attention_heads=2, query_vector_dim=32,
                 batch_size=3, history_len=40, n_candidates=2, n_words_title=40), 2),       # masked 64 x 64 units (d_k = 50) in both encoders
])
def test_v1_model_forward_backward_vs_oracle(shape, title_heads, mode):
    from oracle import nrms_oracle as orc
    SCORE_TOL = TOL[mode]["score"]
    params = synth.make_params_v1(shape, seed=71)
    batch = synth.make_batch(shape, seed=72, ragged=True, min_title=1, mask_some_candidates=True)
    model = make_v1(shape, params, title_heads, precision=mode).train()
    assert len(list(model.parameters())) == 23
    scores, loss, grads = fwd_bwd(model, batch)
    o_scores, o_loss, o_grads, _ = orc.loss_and_grads(orc.v1_to_v0_names(params), batch, shape.num_attention_heads,
                                                      news_heads=title_heads, embed_dropout=False)
    np.testing.assert_allclose(scores, o_scores, rtol=0, atol=SCORE_TOL)
    assert abs(loss - o_loss) < SCORE_TOL
    o_grads = {k: v for k, v in o_grads.items()}
    back = {v: k for k, v in zip(params.keys(), orc.v1_to_v0_names(params).keys())}
    for v0name, g in o_grads.items():
        assert_grad_close(grads[back[v0name]], g, mode, v0name)


@pytest.mark.parametrize("mode", MODES)
def test_v1_dropout_only_after_attention_replayed(mode):
    from oracle import nrms_oracle as orc
    SCORE_TOL = TOL[mode]["score"]
    shape = synth.Shape(n_words=300, word_embed_size=60, num_attention_heads=6, query_vector_dim=32,
                        batch_size=5, history_len=8, n_candidates=3, n_words_title=10)
    params = synth.make_params_v1(shape, seed=81)
    batch = synth.make_batch(shape, seed=82, ragged=True, min_title=2)
    model = make_v1(shape, params, title_heads=3, dropout=0.25, precision=mode).train()
    scores, loss, grads = fwd_bwd(model, batch)
    sv = model.engine._saved
    assert sv["p_embed"] == 0.0 and sv["p"] == 0.25
    n_titles = shape.batch_size * (shape.history_len + shape.n_candidates)
    keep_ctx = model.engine.dropout_keep_mask(sv["seed"], 1, n_titles * shape.n_words_title, 0.25).cpu().view(
        n_titles, shape.n_words_title, shape.word_embed_size)
    o_scores, o_loss, o_grads, _ = orc.loss_and_grads(orc.v1_to_v0_names(params), batch, shape.num_attention_heads,
                                                      p_drop=0.25, keep={"ctx": keep_ctx}, news_heads=3,
                                                      embed_dropout=False)
    np.testing.assert_allclose(scores, o_scores, rtol=0, atol=SCORE_TOL)
    back = {v: k for k, v in zip(params.keys(), orc.v1_to_v0_names(params).keys())}
    for v0name, g in o_grads.items():
        assert_grad_close(grads[back[v0name]], g, mode, v0name)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("mask_mode", [1, 2, 3])
@pytest.mark.parametrize("geom", [(11, 300, 6), (33, 204, 6), (64, 128, 2)], ids=["S11_dk50", "S33_dk34", "S64_dk64"])
def test_masked_primitives_forward_backward(mask_mode, mode, geom):
    """User encoder with v1's masks (UserEncoder.forward(news_vectors, attn_masks), nrms_v1.py:208-211):
    pairwise attention mask (bit 0) and masked additive attention (bit 1), including a fully masked
    sequence (uniform attention over its real positions, as masked_fill(-1e9) gives).  Geometries: a 32 x 64 unit, and
    the edges of the two-wave 64 x 64 units (one query row in the second block / d_k just above 32; everything full)."""
    from oracle import nrms_oracle as orc
    H, d, heads = geom
    shape = synth.Shape(n_words=50, word_embed_size=d, num_attention_heads=heads, query_vector_dim=200,
                        batch_size=5, history_len=H, n_candidates=2, n_words_title=4)
    params = synth.make_params_v1(shape, seed=31)
    SCORE_TOL = TOL[mode]["score"]
    model = make_v1(shape, params, title_heads=heads, precision=mode)
    eng, flat = model.engine, model._flat
    rng = np.random.default_rng(5)
    X = rng.normal(0, 0.5, size=(5, H, d)).astype(np.float32)
    lens = np.array([H, (2 * H) // 3, 1, 0, H - 2])
    mask = (np.arange(H)[None, :] < lens[:, None]).astype(np.uint8)
    dout = rng.normal(0, 1, size=(5, d)).astype(np.float32)
    Xd, md, dd = (torch.from_numpy(a).cuda() for a in (X, mask, dout))
    out = eng.encode_users(flat, Xd, save=True, mask=md, mask_mode=mask_mode)
    gflat = torch.zeros_like(flat)
    dx = eng.encode_users_backward(flat, gflat, Xd, dd, mask=md, mask_mode=mask_mode)
    # oracle
    p = orc.to_torch(orc.v1_to_v0_names(params), requires_grad=True)
    Xt = torch.from_numpy(X).requires_grad_(True)
    o = orc.user_encoder(p, Xt, shape.num_attention_heads, mask=torch.from_numpy(mask), mask_mode=mask_mode)
    (o * torch.from_numpy(dout)).sum().backward()
    np.testing.assert_allclose(out.cpu().numpy(), o.detach().numpy(), rtol=0, atol=SCORE_TOL)
    assert_grad_close(dx.cpu().numpy(), Xt.grad.numpy(), mode, "dx")
    back = {v: k for k, v in zip(params.keys(), orc.v1_to_v0_names(params).keys())}
    for v0name, t in p.items():
        if not v0name.startswith("user_encoder") or t.grad is None:
            continue
        got = model._layout.view(gflat, back[v0name]).cpu().numpy()
        if v0name.endswith("W_K.bias"):
            # sum_i dS[i][j] over the queries: exactly zero without a mask (softmax rows sum to one), what the mask leaves
            # of it is a cancelling sum whose terms are those of d(b_Q) -- bound it by that tensor's scale
            qb = p[v0name.replace("W_K.bias", "W_Q.bias")].grad.numpy()
            tol = TOL[mode]
            assert np.abs(got - t.grad.numpy()).max() <= tol["g_atol"] + (tol["g_rtol"] + tol["g_scale"]) * np.abs(qb).max(), v0name
            continue
        assert_grad_close(got, t.grad.numpy(), mode, v0name)


@pytest.mark.parametrize("mode", MODES + ["fp16", "fp16+user16"])
def test_helper_streams_do_not_change_a_bit(mode, monkeypatch):
    """The backward forks its weight-gradient GEMMs onto helper streams (capi.hip side_streams_for: a set per caller
    stream); with NRMS_NO_SIDE_STREAMS everything runs on the caller's stream.  Same kernels, same accumulation
    order into every gradient buffer: the two must agree bit for bit (a missing dependency between the streams would show
    as a difference), and so must two runs of the same configuration -- EVERY gradient tensor is a fixed-order sum in every
    mode (the fp16 kernels' bias sums went through LDS float atomics until round 3; their title lists were filled in the
    order a race resolved).  v1 topology in the fp32 / bf16x3 modes (output projection, token gather), v0 in fp16."""
    fp16_user = mode.endswith("user16")
    mode = mode.split("+")[0]
    # (a vocabulary large enough that no word occurs more than 64 times: csrc/embed.hip sums longer buckets chunk by chunk)
    shape = synth.Shape(n_words=4000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=24, history_len=50, n_candidates=5, n_words_title=20 if mode != "fp16" else 30)
    batch = synth.make_batch(shape, seed=44, ragged=True, min_title=1, mask_some_candidates=True)
    res = []
    for no_side in (False, True, False):
        if no_side:
            monkeypatch.setenv("NRMS_NO_SIDE_STREAMS", "1")
        else:
            monkeypatch.delenv("NRMS_NO_SIDE_STREAMS", raising=False)
        if mode == "fp16":
            from tests.test_hip_parity import make_model
            model = make_model(shape, synth.make_params(shape, seed=43), precision="fp16", fp16_user=fp16_user).train()
        else:
            model = make_v1(shape, synth.make_params_v1(shape, seed=43), 6, precision=mode).train()
        res.append(fwd_bwd(model, batch))
        torch.cuda.synchronize()
    for other in (res[1], res[2]):
        assert np.array_equal(res[0][0], other[0])
        for n in res[0][2]:
            a, b = res[0][2][n], other[2][n]
            assert np.array_equal(a, b), n
