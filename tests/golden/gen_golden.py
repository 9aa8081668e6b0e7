#!/usr/bin/env python3
"""Generate the golden fixtures by running the IMPORTED REFERENCE on CPU.

Run only in the build container (it needs /root/reference, which never travels to
the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Inputs and weights come from ``pytorch_news_recommender_amd.synth`` (numpy PCG64,
regenerated identically by the tests), so each ``.npz`` stores only OUTPUTS of the
reference's own classes:

  g1_odd.npz    nrms_v0 on an awkward shape (V=97,d=60,h=10,q=32,B=3,H=7,C=3,L=9): ragged
                titles, an all-pad title, an empty-history user, masked candidates, non-zero
                pad row.  news/user vectors, scores, loss, all 19 gradients.
  g2_mind.npz   nrms_v0 at MIND shape (V=2000,d=300,h=10,q=200,B=4,H=50,C=5,L=30): scores,
                loss, vectors, the 18 small gradients + 64 sampled embedding-gradient rows.
  g3_v1.npz     nrms_v1 primitives: MHSA with W_O and pairwise mask, additive attention with
                mask (h=6, d_k=50).
  g4_auc.npz    evaluation.auc_score per padded impression + the mean (train_eval.py:219-271).
  g5_adam.npz   nrms_v0 + torch.optim.Adam(lr=1e-3) for 3 steps, dropout=0: losses and the
                parameters afterwards (64 sampled table rows).
  g6_dataset.npz  the batch-dict items of the reference's own ``MyDataset.__getitem__``
                (data_handler.py:185-250; ``np.int = int`` shim for numpy 2, stub ``nltk``) on the
                hand-written samples of ``synth.dataset_fixture_inputs()``: all 13 keys, train (type 0)
                and evaluation (type 1) padding.

  g7_naml.npz   nrms_naml (model/nrms_naml.py) with dropout 0: an awkward small shape (all 27 gradients in full)
                and the real widths (d=300, 6 heads, q=200, 100-wide category embeddings, 800-wide user encoder with 8
                heads, q=400; B=2): scores, loss, news / user vectors, small gradients in full, the large matrices
                as 32 sampled rows + row sums + column sums.

The reference is imported, never copied; no reference source text is written anywhere.
"""
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/MIND_2020"

from pytorch_news_recommender_amd import synth  # noqa: E402


class cpu_device_patch:
    """nrms_v0.Model.forward hard-codes torch.device('cuda') (nrms_v0.py:248,250,272); in
    this GPU-less container make that resolve to cpu for the duration of the call."""

    def __enter__(self):
        self.orig = torch.device
        orig = self.orig
        torch.device = lambda *a, **k: orig("cpu")
        return self

    def __exit__(self, *exc):
        torch.device = self.orig


def ref_config(shape, tmpdir, table, dropout):
    from config import Config
    np.savez(os.path.join(tmpdir, "all_word_embedding_v3.npz"), embeddings=table)
    c = Config("nrms_v0")
    c.__nrms__()
    c.data_path = tmpdir + "/"
    c.device = torch.device("cpu")
    c.word_embed_size = shape.word_embed_size
    c.num_attention_heads = shape.num_attention_heads
    c.query_vector_dim = shape.query_vector_dim
    c.dropout = dropout
    return c


def build_ref_model(shape, params, dropout=0.0):
    import importlib
    mod = importlib.import_module("model.nrms_v0")
    with tempfile.TemporaryDirectory() as td:
        cfg = ref_config(shape, td, params["news_encoder.word_embedding.0.weight"], dropout)
        model = mod.Model(cfg)
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model


def torch_batch(batch):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}


def run_v0(shape, params, batch, train_mode_no_dropout=True):
    model = build_ref_model(shape, params, dropout=0.0)
    model.train() if train_mode_no_dropout else model.eval()
    tb = torch_batch(batch)
    with cpu_device_patch():
        scores = model(tb)
    loss = torch.nn.CrossEntropyLoss()(scores, torch.zeros(len(scores)).long())
    model.zero_grad()
    loss.backward()
    grads = {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}
    # intermediate vectors through the public helper API (nrms_v0.py:278-299)
    B, H, L = batch["browsed_titles"].shape
    C = batch["candidate_titles"].shape[1]
    with torch.no_grad():
        hist = model.get_news_vector(tb["browsed_titles"].reshape(B * H, L)).view(B, H, -1)
        cand = model.get_news_vector(tb["candidate_titles"].reshape(B * C, L)).view(B, C, -1)
        user = model.get_user_vector(hist)
    return (scores.detach().numpy(), float(loss), grads,
            hist.numpy(), cand.numpy(), user.numpy())


def gen_g1():
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=11, pad_row_zero=False)
    batch = synth.make_batch(shape, seed=12, ragged=True, min_title=1, empty_history_user=True,
                             all_pad_title=True, mask_some_candidates=True)
    scores, loss, grads, hist, cand, user = run_v0(shape, params, batch)
    out = {"scores": scores, "loss": np.float64(loss), "hist": hist, "cand": cand, "user": user}
    for n, g in grads.items():
        out["grad/" + n] = g
    np.savez_compressed(os.path.join(HERE, "g1_odd.npz"), **out)
    print("g1", scores.shape, loss)


def sampled_rows(V, k=64, seed=99):
    return np.sort(np.random.default_rng(seed).choice(V, size=k, replace=False))


def gen_g2():
    shape = synth.G2_MIND
    params = synth.make_params(shape, seed=21)
    batch = synth.make_batch(shape, seed=22, ragged=True)
    scores, loss, grads, hist, cand, user = run_v0(shape, params, batch)
    rows = sampled_rows(shape.n_words)
    rows[0] = 0
    out = {"scores": scores, "loss": np.float64(loss), "user": user,
           "hist": hist.astype(np.float32), "cand": cand.astype(np.float32), "rows": rows}
    for n, g in grads.items():
        if n.endswith("word_embedding.0.weight"):
            out["grad_rows/" + n] = g[rows]
            out["grad_rowsum/" + n] = g.sum(axis=1)          # one float per table row: full coverage
        else:
            out["grad/" + n] = g
    np.savez_compressed(os.path.join(HERE, "g2_mind.npz"), **out)
    print("g2", scores, loss)


def gen_g3():
    """v1 semantics from the reference's own classes (nrms_v1.py:15-105).  nrms_v1 imports
    ``torchsnooper`` (absent here, and only referenced in comments): give it an empty stub."""
    import importlib
    sys.modules.setdefault("torchsnooper", types.ModuleType("torchsnooper"))
    v1 = importlib.import_module("model.nrms_v1")
    rng = np.random.default_rng(31)
    N, S, d, h, q = 5, 11, 300, 6, 200
    X = rng.normal(0, 0.5, size=(N, S, d)).astype(np.float32)
    lens = np.array([11, 7, 1, 4, 9])
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.uint8)
    mh = v1.MultiHeadSelfAttention(h, d, 0.0)
    names = ["W_Q", "W_K", "W_V"]
    sd = {}
    Ws = {}
    for i, n in enumerate(names):
        Ws[n + ".weight"] = rng.uniform(-0.1, 0.1, size=(d, d)).astype(np.float32)
        Ws[n + ".bias"] = rng.uniform(-0.05, 0.05, size=(d,)).astype(np.float32)
        sd["linear_layers.%d.weight" % i] = torch.from_numpy(Ws[n + ".weight"])
        sd["linear_layers.%d.bias" % i] = torch.from_numpy(Ws[n + ".bias"])
    Ws["W_O.weight"] = rng.uniform(-0.1, 0.1, size=(d, d)).astype(np.float32)
    Ws["W_O.bias"] = rng.uniform(-0.05, 0.05, size=(d,)).astype(np.float32)
    sd["output_linear.weight"] = torch.from_numpy(Ws["W_O.weight"])
    sd["output_linear.bias"] = torch.from_numpy(Ws["W_O.bias"])
    mh.load_state_dict(sd)
    Xt = torch.from_numpy(X)
    mt = torch.from_numpy(mask)
    with torch.no_grad():
        y_nomask = mh(Xt, Xt, Xt).numpy()
        y_mask = mh(Xt, Xt, Xt, mask=mt).numpy()
    add = v1.AdditiveAttention(q, d)
    Wa = rng.uniform(-0.1, 0.1, size=(q, d)).astype(np.float32)
    ba = rng.uniform(-0.05, 0.05, size=(q,)).astype(np.float32)
    qv = rng.uniform(-0.1, 0.1, size=(q,)).astype(np.float32)
    add.load_state_dict({"linear.weight": torch.from_numpy(Wa), "linear.bias": torch.from_numpy(ba),
                         "query_vector": torch.from_numpy(qv)})
    with torch.no_grad():
        a_nomask = add(Xt).numpy()
        a_mask = add(Xt, mt).numpy()
    np.savez_compressed(os.path.join(HERE, "g3_v1.npz"), mhsa_nomask=y_nomask, mhsa_mask=y_mask,
                        add_nomask=a_nomask, add_mask=a_mask)
    print("g3", y_mask.shape, a_mask.shape)


def gen_g4():
    ev = __import__("evaluation")
    scores, labels = synth.make_eval_impressions(n_imp=40, max_cand=300, seed=7)
    aucs = np.array([ev.auc_score(y, scores[i][:len(y)]) for i, y in enumerate(labels)])
    np.savez_compressed(os.path.join(HERE, "g4_auc.npz"), aucs=aucs, mean=np.float64(aucs.mean()))
    print("g4", aucs.mean())


def gen_g5():
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=51)
    model = build_ref_model(shape, params, dropout=0.0)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    losses = []
    for t in range(3):
        batch = synth.make_batch(shape, seed=52 + t, ragged=True, min_title=1)
        with cpu_device_patch():
            out = model(torch_batch(batch))
        model.zero_grad()
        loss = crit(out, torch.zeros(len(out)).long())
        losses.append(float(loss))
        loss.backward()
        opt.step()
    outd = {"losses": np.array(losses)}
    for n, p in model.named_parameters():
        outd["param/" + n] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g5_adam.npz"), **outd)
    print("g5", losses)


def gen_g6():
    """MyDataset.__getitem__ of the imported reference on hand-written samples."""
    import pickle
    if "nltk" not in sys.modules:                       # data_handler -> data_processor imports nltk.tokenize;
        nltk = types.ModuleType("nltk")                 # neither is touched by MyDataset
        tok = types.ModuleType("nltk.tokenize")
        tok.word_tokenize = lambda s: s.split()
        tok.RegexpTokenizer = type("RegexpTokenizer", (), {})
        nltk.tokenize = tok
        sys.modules["nltk"], sys.modules["nltk.tokenize"] = nltk, tok
    if not hasattr(np, "int"):
        np.int = int                                    # data_handler.py:191-204 predates numpy 1.24
    import data_handler as ref_dh
    from config import Config
    fx = synth.dataset_fixture_inputs()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "news_title.pkl"), "wb") as f:
            pickle.dump(fx["id2title_dict"], f)
        with open(os.path.join(td, "news_abst.pkl"), "wb") as f:
            pickle.dump(fx["id2abst_dict"], f)
        cfg = Config("g6")
        cfg.data_path = td + "/"
        cfg.mode = "large"
        for k, v in fx["config"].items():
            setattr(cfg, k, v)
        for typ, samples in ((0, fx["train_samples"]), (1, fx["eval_samples"])):
            ds = ref_dh.MyDataset(cfg, samples, type=typ)
            assert len(ds) == len(samples)
            for i in range(len(samples)):
                item = ds[i]
                assert len(item) == 13
                for key, val in item.items():
                    arr = val.numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
                    out["type%d/%d/%s" % (typ, i, key)] = arr
    np.savez_compressed(os.path.join(HERE, "g6_dataset.npz"), **out)
    print("g6", len(out), "arrays")


def naml_sample_rows(n_rows, k=32, seed=77):
    return np.sort(np.random.default_rng(seed).choice(n_rows, size=min(k, n_rows), replace=False))


def gen_g7():
    """nrms_naml.Model of the imported reference (SURVEY f-3)."""
    import importlib
    sys.modules.setdefault("torchsnooper", types.ModuleType("torchsnooper"))    # imported at nrms_naml.py:5, never used
    mod = importlib.import_module("model.nrms_naml")
    from config import Config
    out = {}
    for tag, shape in (("odd", synth.G7_ODD), ("mind", synth.G7_MIND)):
        params = synth.make_params_naml(shape, seed=21)
        batch = synth.make_batch_naml(shape, seed=22)
        with tempfile.TemporaryDirectory() as td:
            np.savez(os.path.join(td, "all_word_embedding_v3.npz"), embeddings=params["news_encoder.word_embedding.weight"])
            cfg = Config("nrms_naml")
            cfg.__nrms__()
            cfg.data_path = td + "/"
            cfg.device = torch.device("cpu")
            cfg.dropout = 0.0
            for k in ("word_embed_size", "title_heads_num", "query_vector_dim", "category_nums", "subcategory_nums",
                      "cate_embed_size", "user_heads_num", "query_vector_dim_large"):
                setattr(cfg, k, getattr(shape, k))
            cfg.news_feature_size = shape.news_feature_size
            model = mod.Model(cfg)
        res = model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        assert list(model.state_dict().keys()) == list(params.keys()), list(model.state_dict().keys())
        model.train()
        tb = torch_batch(batch)
        scores = model(tb)
        loss = torch.nn.CrossEntropyLoss()(scores, torch.zeros(len(scores)).long())
        model.zero_grad()
        loss.backward()
        out[tag + "/scores"] = scores.detach().numpy()
        out[tag + "/loss"] = np.float64(float(loss))
        with torch.no_grad():
            cand = model.news_encoder((tb["candidate_titles"], tb["candidate_absts"], tb["candidate_categ_ids"],
                                       tb["candidate_subcateg_ids"]))
            hist = model.news_encoder((tb["browsed_titles"], tb["browsed_absts"], tb["browsed_categ_ids"],
                                       tb["browsed_subcateg_ids"]))
            user = model.user_encoder(model.norm(hist))
        out[tag + "/cand"], out[tag + "/hist"], out[tag + "/user"] = cand.numpy(), hist.numpy(), user.numpy()
        for name, prm in model.named_parameters():
            g = prm.grad.detach().numpy()
            if g.size <= 100000:
                out[tag + "/grad/" + name] = g.copy()
            else:
                rows = naml_sample_rows(g.shape[0])
                out[tag + "/grad_rows/" + name] = g[rows].copy()
                out[tag + "/grad_rowsum/" + name] = g.sum(1, dtype=np.float64)
                out[tag + "/grad_colsum/" + name] = g.sum(0, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g7_naml.npz"), **out)
    print("g7", len(out), "arrays")


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference not present: fixtures can only be generated in the build container"
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)            # the reference writes nothing, but keep its relative paths away from the repo
        try:
            only = sys.argv[1:]
            for name in ("g1", "g2", "g3", "g4", "g5", "g6", "g7"):
                if not only or name in only:
                    globals()["gen_" + name]()
        finally:
            os.chdir(cwd)
