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
