"""Evaluation path on the GPU: per-impression AUC kernel vs the reference's sklearn values (fixture
g4), unique-title inference vs the plain forward, and the train -> evaluate -> checkpoint loop."""
import os

import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import synth

pytestmark = pytest.mark.gpu


def _model(shape, params, dropout=0.0):
    from tests.test_hip_parity import make_model
    return make_model(shape, params, dropout)


def test_impression_auc_kernel_matches_reference(golden_dir):
    from pytorch_news_recommender_amd import evaluation
    from pytorch_news_recommender_amd.train_eval import _pad_labels
    g = np.load(os.path.join(golden_dir, "g4_auc.npz"))
    scores, labels = synth.make_eval_impressions(n_imp=40, max_cand=300, seed=7)
    model = _model(synth.G1_ODD, synth.make_params(synth.G1_ODD, seed=1))
    eng = model.engine
    dev = torch.device("cuda")
    lab, lens = _pad_labels(labels, 300, dev)
    auc = eng.impression_auc(torch.from_numpy(scores).to(dev), lab, lens).cpu().numpy()
    np.testing.assert_allclose(auc, g["aucs"], rtol=0, atol=1e-12)
    assert abs(auc.mean() - float(g["mean"])) < 1e-12
    # host implementation (same statistic) and the degenerate single-class case
    host = [evaluation.auc_score(y, scores[i][:len(y)]) for i, y in enumerate(labels)]
    np.testing.assert_allclose(host, g["aucs"], rtol=0, atol=1e-12)
    one = eng.impression_auc(torch.zeros(1, 4, device=dev), torch.ones(1, 4, dtype=torch.uint8, device=dev),
                             torch.tensor([4], dtype=torch.int32, device=dev))
    assert torch.isnan(one).all()
    with pytest.raises(ValueError):
        evaluation.auc_score([1, 1], [0.1, 0.2])


def test_unique_title_inference_equals_plain_forward():
    shape = synth.Shape(n_words=50, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=8, history_len=50, n_candidates=40, n_words_title=30)
    params = synth.make_params(shape, seed=3)
    batch = synth.make_batch(shape, seed=4, ragged=True, mask_some_candidates=True)
    # many repeated / all-pad titles, as in real impressions padded to max_candidate_size
    batch["candidate_titles"][:, 10:, :] = 0
    batch["candidate_mask"][:, 10:] = 0
    batch["browsed_titles"][:, 5:, :] = batch["browsed_titles"][:, :1, :]
    model = _model(shape, params).eval()
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    with torch.no_grad():
        model.dedup_inference = True
        s_dedup = model(tb).cpu().numpy()
        n_unique = model.last_unique_titles
        model.dedup_inference = False
        s_plain = model(tb).cpu().numpy()
    assert n_unique < 0.25 * shape.batch_size * (shape.history_len + shape.n_candidates)
    np.testing.assert_allclose(s_dedup, s_plain, rtol=0, atol=1e-6)
    assert (s_dedup[batch["candidate_mask"] == 0] == np.float32(-1e9)).all()


def test_row_grouping_is_exact_and_indexed_scores_match():
    """nrms_title_dedup against torch.unique(dim=0) (same partition of the rows, any numbering) on titles with many
    repeats, near-duplicates (one word apart) and all-padding rows, and on plain ids (seq_len 1);
    nrms_click_score_indexed against nrms_click_score_fwd on the materialised candidates."""
    model = _model(synth.G1_ODD, synth.make_params(synth.G1_ODD, seed=1))
    eng, dev = model.engine, torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    base = torch.randint(1, 40, (300, 30), generator=g)
    near = base[:50].clone()
    near[:, 29] += 1                                                   # differ in the last word only
    rows = torch.cat([base, near, torch.zeros(500, 30, dtype=torch.int64), base[:200], near[:10]])
    rows = rows[torch.randperm(rows.shape[0], generator=g)].contiguous().to(dev)
    for r in (rows, rows[:, :1].contiguous(), rows[:1], rows[:0].reshape(0, 30)):
        inverse, rep, U = eng.group_rows(r)
        ref_u, ref_inv = torch.unique(r, dim=0, return_inverse=True)
        assert U == ref_u.shape[0] == rep.shape[0]
        if r.shape[0] == 0:
            continue
        assert int(inverse.min()) >= 0 and int(inverse.max()) == U - 1
        assert torch.equal(r.index_select(0, rep).index_select(0, inverse), r)          # every row maps to an equal row
        assert torch.unique(torch.stack([inverse.long(), ref_inv], 1), dim=0).shape[0] == U   # one-to-one with torch's groups
    vec = torch.randn(37, 300, generator=g).to(dev)
    user = torch.randn(4, 300, generator=g).to(dev)
    index = torch.randint(0, 37, (4 * 9,), generator=g).to(torch.int32).to(dev)
    mask = (torch.rand(4, 9, generator=g) > 0.3).to(torch.uint8).to(dev)
    want = eng.click_scores(vec.index_select(0, index).view(4, 9, 300), user, mask)
    got = eng.click_scores_indexed(vec, index, user, 4, 9, mask)
    assert torch.equal(got, want)


def test_train_evaluate_checkpoint_loop(tmp_path):
    from torch.utils.data import DataLoader
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.data_handler import MyDataset, SyntheticMind
    from pytorch_news_recommender_amd.model.nrms_hip import Model
    from pytorch_news_recommender_amd import train_eval
    torch.manual_seed(0)
    cfg = Config("nrms_hip")
    cfg.__nrms__()
    cfg.n_words, cfg.n_words_title, cfg.history_len, cfg.sample_size, cfg.max_candidate_size = 600, 12, 10, 4, 24
    cfg.word_embed_size, cfg.num_attention_heads, cfg.query_vector_dim = 60, 6, 32
    cfg.batch_size, cfg.num_epochs, cfg.eval_step, cfg.dropout = 64, 6, 1000, 0.2
    cfg.learning_rate = 4e-3
    cfg.save_path, cfg.log_path = str(tmp_path / "ckpt") + "/", str(tmp_path / "logs")
    corpus = SyntheticMind(cfg, n_news=300, n_topics=4, seed=1)
    model = Model(cfg, pretrained_word_embedding=corpus.embedding_table(cfg.word_embed_size)).to("cuda")
    train_ds = MyDataset(cfg, corpus.train_samples(2048), type=0, id2title_dict=corpus.id2title_dict)
    dev_samples, dev_labels = corpus.eval_samples(256, max_shown=20)
    dev_ds = MyDataset(cfg, dev_samples, type=1, id2title_dict=corpus.id2title_dict)
    item = train_ds[0]
    assert item["browsed_titles"].shape == (10, 12) and item["browsed_titles"].dtype == np.int64
    assert item["candidate_titles"].shape == (5, 12) and item["candidate_mask"].dtype == torch.uint8
    assert dev_ds[0]["candidate_titles"].shape == (24, 12)
    tl = DataLoader(train_ds, batch_size=cfg.batch_size, shuffle=True, num_workers=0)
    dl = DataLoader(dev_ds, batch_size=cfg.batch_size, shuffle=False, num_workers=0)
    auc0 = train_eval.evaluate(cfg, model, dl, dev_labels, verbose=False)
    hist = train_eval.train(cfg, model, tl, dl, dev_labels, verbose=False)
    print("losses", np.round(hist["losses"][::16], 3), "aucs", auc0, hist["aucs"])
    assert np.isfinite(hist["losses"]).all() and len(hist["losses"]) == 6 * 32
    assert np.mean(hist["losses"][-8:]) < np.mean(hist["losses"][:8]) - 0.05       # it learns
    auc1 = hist["aucs"][-1][1]
    assert auc1 > max(auc0, 0.5) + 0.05, (auc0, auc1)
    assert hist["ckpts"], "dev AUC improved past 0.56, a checkpoint must have been written"
    # checkpoint interchange: reference key names, loads back bit-exactly
    sd = torch.load(os.path.join(cfg.save_path, hist["ckpts"][-1]), weights_only=True)
    assert sorted(sd) == sorted(synth.param_names())
    m2 = Model(cfg, pretrained_word_embedding=corpus.embedding_table(cfg.word_embed_size)).to("cuda")
    m2.load_state_dict(sd)
    assert abs(train_eval.evaluate(cfg, m2, dl, dev_labels, verbose=False) - auc1) < 1e-9
    # the evaluation above ran on the persistent news-vector cache (the batch dicts carry news ids): every news
    # item was encoded once for the whole dev set; the plain path (every slot encoded) gives the same AUC
    stats = m2.last_eval_cache
    assert stats["encoded"] <= 301 and stats["lookups"] == 256 * (10 + 24), stats
    m2.dedup_inference = False
    assert abs(train_eval.evaluate(cfg, m2, dl, dev_labels, verbose=False) - auc1) < 1e-6
    m2.dedup_inference = True
    # the same trained weights evaluated under the other precision settings: bf16x3 directly, and "fp16", whose evaluation
    # passes run in bf16x3 by default (config.fp16_inference) -- the dev AUC must agree to north_star's 1e-4 (one rank flip
    # among ~20 candidates of one of these 256 impressions would already be 2e-4)
    for prec in ("bf16x3", "fp16"):
        cfg.precision = prec
        auc_p = train_eval.evaluate(cfg, m2, dl, dev_labels, verbose=False)
        assert m2.engine.precision == prec
        assert abs(auc_p - auc1) <= 1e-4, (prec, auc_p, auc1)
    cfg.precision = "fp32"
    out = train_eval.test(cfg, m2, dl, [len(y) for y in dev_labels], out_file=str(tmp_path / "sub.txt"))
    first = open(out).readline().split(" ", 1)
    assert first[0] == "1" and sorted(eval(first[1])) == list(range(1, len(dev_labels[0]) + 1))
    # the reference's literal loop (autograd + torch.optim.Adam) drives the same kernels
    h2 = train_eval.train(cfg, m2, tl, None, None, use_autograd=True, max_batches=3, verbose=False)
    assert len(h2["losses"]) == 3 and np.isfinite(h2["losses"]).all()


def test_warmup_phase_against_oracle():
    """config.warm_up: the first batches run at the reference's linearly increasing learning rate
    (train_eval.py:64-99); 5 warm-up iterations on the G1 shape against the oracle stepped with the
    same per-iteration rate (iterations 0 and 1 run at lr = 0, so warm_up_steps is shrunk to 4)."""
    from oracle import nrms_oracle as orc
    from pytorch_news_recommender_amd import train_eval
    from tests.test_hip_parity import tbatch, assert_params_close
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=61)
    model = _model(shape, params)
    cfg = model.config
    cfg.warm_up, cfg.warm_up_steps, cfg.learning_rate, cfg.num_epochs = True, 4, 2e-3, 0
    batches = [synth.make_batch(shape, seed=70 + t, ragged=True, min_title=1) for t in range(5)]
    hist = train_eval.train(cfg, model, [tbatch(b) for b in batches], max_batches=5, verbose=False)
    assert len(hist["warmup_losses"]) == 5 and not hist["losses"]
    # oracle with the same schedule
    p = {k: v.astype(np.float32).copy() for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v_ = {k: np.zeros_like(v) for k, v in p.items()}
    ref_losses = []
    for t, b in enumerate(batches):
        _, loss, grads, _ = orc.loss_and_grads(p, b, shape.num_attention_heads)
        ref_losses.append(loss)
        for k in p:
            orc.adam_step(p[k], grads[k].astype(np.float32), m[k], v_[k], t + 1,
                          lr=train_eval.warmup_lr(2e-3, t, 4))
    np.testing.assert_allclose(hist["warmup_losses"], ref_losses, atol=2e-6)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for n in synth.param_names():
        if n.endswith("W_K.bias"):
            continue
        assert_params_close(sd[n], p[n], n)


def test_eval_c300_scores_and_auc_against_oracle():
    """End to end at the evaluation shape (C = max_candidate_size = 300 padded candidate slots, most of them
    padding or repeats: data_handler.py:174-177): batch dict -> Model.forward in eval mode (unique-title path,
    title keys hashed on the device) -> per-impression AUC kernel, against the oracle's scores and its
    rank-statistic AUC on the same impressions."""
    from oracle import nrms_oracle as orc
    from pytorch_news_recommender_amd import train_eval
    shape = synth.Shape(n_words=500, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=6, history_len=50, n_candidates=300, n_words_title=30)
    params = synth.make_params(shape, seed=41)
    rng = np.random.default_rng(42)
    B, H, C, L = 6, 50, 300, 30
    # a small corpus so that titles repeat across users and slots, as real impressions do
    corpus = rng.integers(1, shape.n_words, size=(120, L))
    corpus = np.where(np.arange(L)[None, :] < rng.integers(4, L + 1, size=(120, 1)), corpus, 0)
    hist_len = rng.integers(3, H + 1, size=B)
    shown = rng.integers(8, 60, size=B)
    bt = np.zeros((B, H, L), dtype=np.int64)
    ct = np.zeros((B, C, L), dtype=np.int64)
    cm = np.zeros((B, C), dtype=np.uint8)
    labels = []
    for b in range(B):
        bt[b, :hist_len[b]] = corpus[rng.integers(0, 120, size=hist_len[b])]
        ct[b, :shown[b]] = corpus[rng.choice(120, size=shown[b], replace=False)]
        cm[b, :shown[b]] = 1
        y = (rng.random(shown[b]) < 0.2).astype(np.int64)
        y[0], y[1] = 1, 0
        labels.append(y.tolist())
    batch = {"browsed_titles": bt, "candidate_titles": ct, "candidate_mask": cm}
    model = _model(shape, params).eval()
    with torch.no_grad():
        scores = model({k: torch.from_numpy(v) for k, v in batch.items()}).cpu().numpy()
    assert model.last_unique_titles <= 121 + 1                    # 120 corpus titles + the all-padding title
    p = orc.to_torch(params)
    with torch.no_grad():
        o_scores, _ = orc.forward(p, batch, shape.num_attention_heads)
    o_scores = o_scores.numpy()
    valid = cm == 1
    np.testing.assert_allclose(scores[valid], o_scores[valid], rtol=0, atol=1e-5)
    assert (scores[~valid] == np.float32(-1e9)).all()
    o_mean, o_aucs = orc.mean_impression_auc(o_scores, labels)
    # the comparison is only meaningful where no positive/negative pair is closer than the score tolerance
    for b in range(B):
        s = np.sort(o_scores[b, :shown[b]])
        assert np.diff(s).min() > 2e-5, "near-tie in the oracle scores: pick another seed"
    auc = train_eval.evaluate(model.config, model, [{k: torch.from_numpy(v) for k, v in batch.items()}], labels,
                              verbose=False)
    assert abs(auc - o_mean) <= 1e-4, (auc, o_mean)               # north_star: AUC within 1e-4 (here: identical ranks)
    assert abs(auc - o_mean) <= 1e-12


def test_run_demo_entry_point_batch_32(tmp_path, monkeypatch):
    """BASELINE.json config 0: the demo entry (run_demo -> run_v0.main) at batch 32, through the reference's
    wrapper contract model.Model(config, args): a few training steps, a dev evaluation, finite loss."""
    from pytorch_news_recommender_amd import run_demo, run_v0
    monkeypatch.chdir(tmp_path)
    argv = run_demo.demo_argv(["--synthetic_users", "256", "--epochs", "1", "--max_batches", "6", "--num_workers", "0",
                               "--data_path", str(tmp_path / "data_processed"), "--save_path", str(tmp_path / "save")])
    assert argv[argv.index("--batch_size") + 1] == "32"
    hist = run_v0.main(argv)
    assert len(hist["losses"]) == 6 and np.isfinite(hist["losses"]).all()
    assert hist["aucs"] and 0.0 < hist["aucs"][-1][1] < 1.0
    assert os.path.exists(tmp_path / "data_processed" / "all_word_embedding_v3.npz")


def _rank_flips(s_a, s_b, labels):
    """(positive, negative) pairs of an impression whose order differs between two score sets (ties count as half a flip,
    as they do in the rank statistic)."""
    flips, where = 0.0, []
    for i, y in enumerate(labels):
        y = np.asarray(y, dtype=bool)
        a, b = s_a[i, :len(y)], s_b[i, :len(y)]
        ca = np.sign(a[y][:, None] - a[~y][None, :])
        cb = np.sign(b[y][:, None] - b[~y][None, :])
        f = float(np.abs(ca - cb).sum()) / 2.0
        if f:
            flips += f
            where.append((i, f, int(y.sum()) * int((~y).sum())))
    return flips, where


def test_auc_parity_of_the_fp16_mode_at_the_train_parity_size():
    """north_star: click scores AND AUC within 1e-4 of the reference (evaluate(), train_eval.py:229-273).  The size of
    tools/train_parity.py: bench dimensions, 240 Adam steps of 256 users WITH dropout 0.2 in the exact fp32 mode (so the weights
    are trained ones and the scores are no longer the small scores of an initialisation), then the SAME weights evaluated on
    1 024 dev impressions in every mode.  precision = "fp16" evaluates in bf16x3 by default (config.fp16_inference = False): its
    dev AUC must sit within 1e-4 of the fp32 mode's (measured: identical to ~1e-7), every score within 2e-5.  The opt-in
    fp16_inference = True is what round 3 measured at 1.4e-4: its gap is a handful of rank flips between candidates whose fp32
    scores are closer than the fp16 score error -- counted and printed here, and held to a looser, stated bar (4e-4)."""
    from pytorch_news_recommender_amd import train_eval
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.data_handler import DeviceFeed, SyntheticMind
    from pytorch_news_recommender_amd.model.nrms_hip import Model
    cfg = Config("nrms_hip")
    cfg.__nrms__()
    cfg.n_words_title, cfg.batch_size, cfg.dropout, cfg.precision, cfg.learning_rate = 30, 256, 0.2, "fp32", 1e-3
    cfg.max_candidate_size = 40
    corpus = SyntheticMind(cfg, n_news=4000, seed=0)
    table = torch.from_numpy(np.asarray(corpus.embedding_table(cfg.word_embed_size), dtype=np.float32))
    torch.manual_seed(0)
    model = Model(cfg, pretrained_word_embedding=table).cuda().train()
    feed = DeviceFeed(cfg, corpus.train_samples(256 * 40), type=0, id2title_dict=corpus.id2title_dict,
                      id2abst_dict=corpus.id2abst_dict, batch_size=256, device="cuda", shuffle=True, drop_last=True, seed=3)
    dev_samples, dev_labels = corpus.eval_samples(1024, max_shown=30)
    dev = DeviceFeed(cfg, dev_samples, type=1, id2title_dict=corpus.id2title_dict, id2abst_dict=corpus.id2abst_dict,
                     batch_size=256, device="cuda")
    steps = 0
    while steps < 240:
        for b in feed:
            model.train_step(b)
            steps += 1
            if steps >= 240:
                break
    res = {}
    for tag, prec, inf16 in (("fp32", "fp32", False), ("bf16x3", "bf16x3", False), ("fp16 (default: bf16x3 inference)", "fp16", False),
                             ("fp16 + fp16_inference", "fp16", True)):
        cfg.precision, cfg.fp16_inference = prec, inf16
        auc = train_eval.evaluate(cfg, model, dev, dev_labels, verbose=False)
        res[tag] = (auc, model.last_eval_scores.cpu().numpy().copy())
    cfg.precision, cfg.fp16_inference = "fp32", False
    auc32, s32 = res["fp32"]
    valid = np.zeros_like(s32, dtype=bool)
    for i, y in enumerate(dev_labels):
        valid[i, :len(y)] = True
    print("dev AUC fp32 %.6f over %d impressions, scores rms %.3f max %.3f" % (auc32, len(dev_labels), float(np.sqrt((s32[valid] ** 2).mean())),
                                                                             float(np.abs(s32[valid]).max())))
    for tag, (auc, s) in res.items():
        if tag == "fp32":
            continue
        flips, where = _rank_flips(s32, s, dev_labels)
        err = float(np.abs(s - s32)[valid].max())
        print("  %-34s AUC %.6f  gap %.2e  max |score - fp32| %.2e  rank flips %.1f in impressions %s" % (
            tag, auc, abs(auc - auc32), err, flips, [(i, f, "of %d pairs" % n) for i, f, n in where][:8]))
        if tag == "fp16 + fp16_inference":
            assert abs(auc - auc32) < 4e-4, (tag, auc, auc32)          # opt-in: a few flips of near-tied candidates
        else:
            assert abs(auc - auc32) <= 1e-4 and err < 2e-5 * max(1.0, float(np.abs(s32[valid]).max())), (tag, auc, auc32, err)


def test_fp16_mode_evaluates_in_bf16x3_unless_asked():
    """config.precision = "fp16": passes that keep nothing for a backward are routed to the split-bf16 kernels (1e-6), training
    passes to the fused fp16 kernels; config.fp16_inference = True puts inference on the fp16 kernels too."""
    from pytorch_news_recommender_amd import _lib
    shape = synth.Shape(n_words=500, word_embed_size=300, num_attention_heads=10, query_vector_dim=200, batch_size=4,
                        history_len=50, n_candidates=5, n_words_title=30)
    from tests.test_hip_parity import make_model
    model = make_model(shape, synth.make_params(shape, seed=3), precision="fp16", fp16_inference=False)
    eng = model.engine
    assert eng._desc("news_encoder", 8, 30, training=True).precision == _lib.NRMS_PRECISION_FP16
    assert eng._desc("news_encoder", 8, 30, training=False).precision == _lib.PRECISIONS["bf16x3"]
    assert eng._desc("user_encoder", 4, 50, training=True).precision == _lib.PRECISIONS["bf16x3"]
    model.config.fp16_inference = True
    assert model.engine._desc("news_encoder", 8, 30, training=False).precision == _lib.NRMS_PRECISION_FP16
