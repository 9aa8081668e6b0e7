"""CPU-only checks: the C-ABI library loads and exports every symbol include/nrms_hip.h declares
(no compute call without a GPU), the flat parameter layout, the drop-in Model's names /
state_dict / loud failure without a GPU, and the synthetic batch layout."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import _lib, synth
from pytorch_news_recommender_amd.config import Config
from pytorch_news_recommender_amd.engine import FlatLayout, ModelDims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "nrms_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nrms_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 12
    assert sorted(_lib.SIGNATURES) == names, "ctypes signatures and the header disagree"
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -m pytorch_news_recommender_amd.build"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    bound = _lib.load()
    assert b"gfx950" in bound.nrms_version()
    assert bound.nrms_last_error() is not None


def test_struct_layouts_match_header():
    # field order / widths of the POD structs (x86-64 SysV): desc = 12 x 4 bytes, u64, 2 x 4 bytes, one pointer
    assert ctypes.sizeof(_lib.EncoderDesc) == 72
    assert _lib.EncoderDesc.seed.offset == 48 and _lib.EncoderDesc.loss_scale.offset == 56 and _lib.EncoderDesc.seq_index.offset == 64
    assert ctypes.sizeof(_lib.EncoderWeights) == 64 and ctypes.sizeof(_lib.EncoderGrads) == 64
    assert ctypes.sizeof(_lib.EncoderActs) == 56


def test_argument_validation_without_gpu():
    """Validation runs before any HIP call, so bad descriptors are reported on a CPU-only host."""
    lib = _lib.load()
    bad = _lib.EncoderDesc(n_seq=1, seq_len=65, d_model=300, n_heads=10, q_dim=200, vocab=0, p_drop_embed=0.0,
                           p_drop_ctx=0.0, precision=0, seed=0)
    assert lib.nrms_encoder_bwd_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"seq_len" in lib.nrms_last_error()
    ok = _lib.EncoderDesc(n_seq=28160, seq_len=30, d_model=300, n_heads=10, q_dim=200, vocab=45800,
                          p_drop_embed=0.2, p_drop_ctx=0.2, precision=0, seed=1)
    need = lib.nrms_encoder_bwd_workspace_bytes(ctypes.byref(ok))
    M = 28160 * 30
    assert need >= 4 * (M * 300 + M * 900 + M)           # dctx + dqkv + ds at least
    w = _lib.EncoderWeights()
    acts = _lib.EncoderActs()
    rc = lib.nrms_encoder_fwd(ctypes.byref(ok), ctypes.byref(w), None, None, None, ctypes.byref(acts), None, None)
    assert rc == -1 and b"null" in lib.nrms_last_error()
    odd = _lib.EncoderDesc(n_seq=1, seq_len=5, d_model=30, n_heads=10, q_dim=200, vocab=0, p_drop_embed=0.0,
                           p_drop_ctx=0.0, precision=0, seed=0)
    assert lib.nrms_encoder_bwd_workspace_bytes(ctypes.byref(odd)) == 0      # d_model % 4 != 0
    with pytest.raises(_lib.NrmsError):
        _lib.check(-1, "unit test")


def test_flat_layout_adjacency_and_names():
    dims = ModelDims(n_words=45800, word_embed_size=300, num_attention_heads=10, query_vector_dim=200)
    lay = FlatLayout(dims)
    assert lay.total == 14402600
    assert list(lay.entries) == synth.param_names()
    assert len(lay.entries) == 19
    v1 = FlatLayout(ModelDims(n_words=45800, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                              news_heads=6, output_proj=True, style="v1"))
    assert len(v1.entries) == 23 and v1.total == lay.total + 2 * (300 * 300 + 300)
    assert "user_encoder.multi_head_self_attention.output_linear.weight" in v1.entries
    assert "news_encoder.additive_attention.query_vector" in v1.entries
    for off, shp, n in lay.entries.values():
        assert off % 4 == 0 and n == int(np.prod(shp))
    flat = torch.arange(lay.total, dtype=torch.float32)
    v = lay.view(flat, "user_encoder.multihead_self_attention.W_K.weight")
    assert v.shape == (300, 300) and v.data_ptr() == flat.data_ptr() + 4 * lay.entries[
        "user_encoder.multihead_self_attention.W_K.weight"][0]


def make_cfg(shape):
    cfg = Config("nrms_hip")
    cfg.__nrms__()
    cfg.word_embed_size = shape.word_embed_size
    cfg.num_attention_heads = shape.num_attention_heads
    cfg.query_vector_dim = shape.query_vector_dim
    return cfg


def test_model_names_state_dict_and_loud_failure_on_cpu(tmp_path):
    from pytorch_news_recommender_amd.model.nrms_hip import Model
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=3)
    cfg = make_cfg(shape)
    # the reference's own construction path: table from config.data_path + npz (nrms_v0.py:134-135)
    np.savez(tmp_path / "all_word_embedding_v3.npz", embeddings=params["news_encoder.word_embedding.0.weight"])
    cfg.data_path = str(tmp_path) + "/"
    m = Model(cfg)
    assert sorted(m.state_dict()) == sorted(synth.param_names())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    assert m._views_intact()
    for n, p in m.named_parameters():
        np.testing.assert_array_equal(p.detach().numpy(), params[n])
    np.testing.assert_array_equal(m._layout.view(m._flat, "news_encoder.additive_attention.linear.bias").numpy(),
                                  params["news_encoder.additive_attention.linear.bias"])
    # module protocol used by train_eval.py: train/eval/parameters/zero_grad/state_dict
    m.train(); m.eval(); m.zero_grad()
    assert sum(p.numel() for p in m.parameters()) == m._layout.total
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(shape, seed=4).items()}
    with pytest.raises(_lib.NrmsError, match="no CPU fallback"):
        m(batch)                                          # never a silent CPU path
    with pytest.raises(_lib.NrmsError):
        m.get_news_vector(batch["browsed_titles"][0])


def test_model_dispatch_wrapper():
    import types
    from pytorch_news_recommender_amd import model as model_pkg
    shape = synth.G1_ODD
    cfg = make_cfg(shape)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "all_word_embedding_v3.npz"),
                 embeddings=synth.make_params(shape, seed=3)["news_encoder.word_embedding.0.weight"])
        cfg.data_path = td + "/"
        args = types.SimpleNamespace(model="NRMS_V0", n_GPUs=1)
        w = model_pkg.Model(cfg, args)
    assert sorted(w.state_dict()) == sorted("model." + n for n in synth.param_names())   # ckpt keys, SURVEY 5


def test_config_fields_of_the_reference():
    c = Config("x")
    c.__nrms__()
    for f in ("data_path", "word_embedding_pretrained", "word_embed_size", "num_attention_heads",
              "title_heads_num", "query_vector_dim", "dropout", "history_len", "sample_size",
              "max_candidate_size", "n_words_title", "batch_size", "learning_rate", "device", "eval_step",
              "num_epochs", "save_path", "log_path", "warm_up", "warm_up_steps"):
        assert hasattr(c, f), f
    assert (c.n_words, c.word_embed_size, c.history_len, c.sample_size, c.dropout, c.learning_rate) == (
        45800, 300, 50, 5, 0.2, 1e-3)
    assert (c.query_vector_dim, c.num_attention_heads, c.title_heads_num) == (200, 10, 6)


def test_synthetic_batch_layout():
    shape = synth.G1_ODD
    b = synth.make_batch(shape, seed=12, ragged=True, min_title=1, empty_history_user=True,
                         all_pad_title=True, mask_some_candidates=True)
    B, H, C, L = shape.batch_size, shape.history_len, shape.n_candidates, shape.n_words_title
    assert b["browsed_titles"].shape == (B, H, L) and b["browsed_titles"].dtype == np.int64
    assert b["candidate_titles"].shape == (B, C, L) and b["candidate_mask"].dtype == np.uint8
    assert b["browsed_mask"].dtype == np.uint8
    # left-aligned history, right-zero-padded titles (data_handler.py:206-215)
    for u in range(B):
        n = int(b["browsed_lens"][u])
        assert not b["browsed_titles"][u, n:].any()
        assert b["browsed_mask"][u, :n].all() and not b["browsed_mask"][u, n:].any()
    t = b["candidate_titles"]
    nz = t != 0
    assert (np.diff(nz.astype(int), axis=-1) <= 0).all()          # once padded, stays padded
    assert t.max() < shape.n_words


def test_warmup_schedule_matches_reference_scheduler():
    """train_eval.py:64-99 + lr_scheduler.py:41-42: linear ramp from 0, stepped after each iteration."""
    import torch
    from pytorch_news_recommender_amd.train_eval import warmup_lr, warmup_iterations

    class Ramp(torch.optim.lr_scheduler.LRScheduler):          # the reference's multiplier == 1 branch
        def __init__(self, opt, total):
            self.total = total
            super().__init__(opt)

        def get_lr(self):
            return [b * min(float(self.last_epoch) / self.total, 1.0) for b in self.base_lrs]

    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    sch = Ramp(opt, 500)
    opt.zero_grad(); opt.step()
    used = []
    for i in range(warmup_iterations(500)):
        used.append(opt.param_groups[0]['lr'])
        p.grad = torch.ones(1); opt.step()
        if i > 500:
            break
        sch.last_epoch = i - 1; sch.step()                      # == scheduler.step(i) of the reference
    assert len(used) == 502
    for i, lr in enumerate(used):
        assert abs(lr - warmup_lr(1e-3, i, 500)) < 1e-12, (i, lr)
    assert used[0] == 0.0 and used[1] == 0.0 and abs(used[501] - 1e-3) < 1e-15


def test_engine_scratch_bound_covers_library_sizes_and_flag_validation():
    """The host side sizes the forward scratch with a closed-form bound (engine._acts); it must cover
    nrms_encoder_fwd_scratch_bytes for every mode, with and without NRMS_FLAG_PAD_ROW_ZERO; unknown flag
    bits are rejected."""
    lib = _lib.load()
    for d, h, q, n_seq, S in [(300, 10, 200, 28160, 30), (512, 8, 256, 1000, 64), (60, 6, 32, 7, 9), (8, 2, 4, 1, 1)]:
        M = n_seq * S
        bound_floats = 3 * d * d + 3 * d + 256 + (3 * d + 32) * (d + 32) + (M + M // 1024 + 512)
        for prec in (0, 1, 2):
            for vocab, flags in ((0, 0), (100, 0), (100, _lib.NRMS_FLAG_PAD_ROW_ZERO)):
                desc = _lib.EncoderDesc(n_seq=n_seq, seq_len=S, d_model=d, n_heads=h, q_dim=q, vocab=vocab,
                                        p_drop_embed=0.0, p_drop_ctx=0.0, precision=prec, flags=flags, seed=0)
                need = lib.nrms_encoder_fwd_scratch_bytes(ctypes.byref(desc))
                assert 0 < need <= 4 * bound_floats, (d, prec, vocab, flags, need, 4 * bound_floats)
                assert lib.nrms_encoder_bwd_workspace_bytes(ctypes.byref(desc)) > 0
    bad = _lib.EncoderDesc(n_seq=1, seq_len=5, d_model=60, n_heads=6, q_dim=32, vocab=10, p_drop_embed=0.0,
                           p_drop_ctx=0.0, precision=0, flags=64, seed=0)
    assert lib.nrms_encoder_fwd_scratch_bytes(ctypes.byref(bad)) == 0
    assert b"flags" in lib.nrms_last_error()


def test_fp16_descriptor_rules_without_gpu():
    """What precision fp16 accepts is decided on the host (validate_desc): v0's fused kernels (no W_O, d_k <= 32, h <= 10), and with
    use_output_proj the nrms_v1 news encoder (32 < d_k <= 50, padding-skipping path, no embedding dropout, seq_len <= 32).
    The sizing calls answer 0 + an error text for everything else -- the host driver routes those passes to bf16x3."""
    lib = _lib.load()
    base = dict(n_seq=64, seq_len=20, d_model=300, n_heads=6, q_dim=200, vocab=1000, p_drop_embed=0.0, p_drop_ctx=0.2, precision=3,
                use_output_proj=1, mask_mode=0, flags=_lib.NRMS_FLAG_PAD_ROW_ZERO, seed=0, loss_scale=0.0, p_drop_attn=0.0)
    ok = _lib.EncoderDesc(**base)
    fwd, bwd = lib.nrms_encoder_fwd_scratch_bytes(ctypes.byref(ok)), lib.nrms_encoder_bwd_workspace_bytes(ctypes.byref(ok))
    assert fwd > 0 and bwd > 64 * 32 * 320 * 2 * 3, (fwd, bwd)            # (dZ16, d(ctx)16, dQKV16 ... of 64 titles)
    for kw, word in ((dict(mask_mode=1), b"mask"), (dict(seq_len=40), b"seq_len"), (dict(n_heads=10), b"d_k"), (dict(flags=0), b"PAD_ROW_ZERO"),
                     (dict(p_drop_embed=0.1), b"embedding dropout"), (dict(vocab=0, flags=0), b"news encoder"), (dict(d_model=320, n_heads=8), b"d_model"),
                     (dict(p_drop_attn=0.1), b"p_drop_attn")):
        bad = _lib.EncoderDesc(**dict(base, **kw))
        assert lib.nrms_encoder_fwd_scratch_bytes(ctypes.byref(bad)) == 0, kw
        msg = lib.nrms_last_error()
        assert b"fp16" in msg and word in msg, (kw, msg)
    v0 = _lib.EncoderDesc(**dict(base, use_output_proj=0, n_heads=10, seq_len=30))
    assert lib.nrms_encoder_fwd_scratch_bytes(ctypes.byref(v0)) > 0
    wide = _lib.EncoderDesc(**dict(base, use_output_proj=0))                     # six heads of 50 without W_O: not a fused shape
    assert lib.nrms_encoder_fwd_scratch_bytes(ctypes.byref(wide)) == 0 and b"d_k <= 32" in lib.nrms_last_error()


# ---- data feed (SURVEY a-12, f-2) -------------------------------------------------------------------
def _g6_cfg(tmp_path=None):
    fx = synth.dataset_fixture_inputs()
    cfg = Config("g6")
    for k, v in fx["config"].items():
        setattr(cfg, k, v)
    if tmp_path is not None:
        cfg.data_path = str(tmp_path) + "/"
    return fx, cfg


def _check_against_g6(ds_by_type, golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_dataset.npz"), allow_pickle=False)
    keys = sorted({k.split("/", 2)[2] for k in g.files})
    assert len(keys) == 13
    for typ, ds in ds_by_type.items():
        for i in range(len(ds)):
            item = ds[i]
            assert sorted(item) == keys
            for key in keys:
                want = g["type%d/%d/%s" % (typ, i, key)]
                got = item[key].numpy() if isinstance(item[key], torch.Tensor) else np.asarray(item[key])
                assert got.dtype == want.dtype, (typ, i, key, got.dtype, want.dtype)
                np.testing.assert_array_equal(got, want, err_msg="type%d/%d/%s" % (typ, i, key))
            assert isinstance(item["browsed_mask"], torch.Tensor) and item["browsed_mask"].dtype == torch.uint8


def test_mydataset_items_equal_the_reference_dataset(golden_dir):
    """The 13 arrays per item (keys, dtypes, padding, truncation to sample_size+1 / max_candidate_size) against
    what the IMPORTED reference MyDataset.__getitem__ produced on the same hand-written samples (fixture g6,
    tests/golden/gen_golden.py::gen_g6; data_handler.py:185-250)."""
    from pytorch_news_recommender_amd.data_handler import MyDataset
    fx, cfg = _g6_cfg()
    ds = {0: MyDataset(cfg, fx["train_samples"], type=0, id2title_dict=fx["id2title_dict"], id2abst_dict=fx["id2abst_dict"]),
          1: MyDataset(cfg, fx["eval_samples"], type=1, id2title_dict=fx["id2title_dict"], id2abst_dict=fx["id2abst_dict"])}
    assert ds[0].sample_size == 3 and ds[1].sample_size == 6
    _check_against_g6(ds, golden_dir)


def test_reference_data_files_are_read_like_the_reference(tmp_path, golden_dir):
    """The reference's constructor signature MyDataset(config, datas, type) with the word dictionaries coming from
    config.data_path: news_words.csv (headerless news_id,title,abstract list literals) -> cached news_title.pkl /
    news_abst.pkl (data_handler.py:113-135); dev labels from dev_behaviors.csv:y_true (train_eval.py:36-39);
    idx_<file> sample cache (data_handler.py:43-47); test list lengths (train_eval.py:287-298); best checkpoint
    by file-name AUC (train_eval.py:301-308)."""
    import csv
    import pickle
    from pytorch_news_recommender_amd import data_handler as dh, train_eval
    fx, cfg = _g6_cfg(tmp_path)
    with open(tmp_path / "news_words.csv", "w", newline="") as f:
        w = csv.writer(f)
        for i in range(6):
            w.writerow(["N%d" % (i + 1), str(fx["id2title_dict"][i]), str(fx["id2abst_dict"][i])])
    ds = {0: dh.MyDataset(cfg, fx["train_samples"], type=0), 1: dh.MyDataset(cfg, fx["eval_samples"], 1)}
    assert os.path.exists(tmp_path / "news_title.pkl") and os.path.exists(tmp_path / "news_abst.pkl")
    _check_against_g6(ds, golden_dir)
    _check_against_g6({0: dh.MyDataset(cfg, fx["train_samples"])}, golden_dir)          # now from the pickles
    with open(tmp_path / "dev_behaviors.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["user_id", "time", "history", "impressions", "y_true"])
        w.writerow(["U1", "t", "N1 N2", "N3-1 N4-0", "1 0"])
        w.writerow(["U2", "t", "N2", "N1-0 N5-0 N6-1", "0 0 1"])
    assert dh.read_dev_labels(cfg) == [[1, 0], [0, 0, 1]]
    with open(tmp_path / "idx_train_datas.pkl", "wb") as f:
        pickle.dump(fx["train_samples"], f)
    assert dh.load_dataset(cfg, "train_datas.pkl", cfg.data_path, _type=0) == fx["train_samples"]
    with pytest.raises(FileNotFoundError):
        dh.load_dataset(cfg, "dev_datas.pkl", cfg.data_path, _type=1)
    os.makedirs(tmp_path / "test")
    cfg.test_path = str(tmp_path / "test") + "/"
    with open(tmp_path / "test" / "behaviors.tsv", "w") as f:
        f.write("1\tU1\tt\tN1 N2\tN3 N4 N5\n2\tU2\tt\tN2\tN1\n")
    assert dh.get_Test_List(cfg) == [3, 1] and os.path.exists(tmp_path / "test_imps_list.pkl")
    cfg.save_path = str(tmp_path / "save") + "/"
    assert train_eval.best_checkpoint(cfg) is None
    os.makedirs(cfg.save_path)
    for name in ("T10-01_10.00_g6_epoch6_iter_100_auc_0.612.ckpt", "T10-01_11.00_g6_epoch6_iter_200_auc_0.655.ckpt",
                 "T10-01_12.00_other_epoch6_iter_300_auc_0.700.ckpt", "T10-01_13.00_g6_epoch6_iter_400_auc_0.640.ckpt"):
        open(os.path.join(cfg.save_path, name), "w").close()
    assert train_eval.best_checkpoint(cfg) == "T10-01_11.00_g6_epoch6_iter_200_auc_0.655.ckpt"


def test_loading_through_the_wrapper_resets_the_padding_row_flag(tmp_path):
    """NRMS_FLAG_PAD_ROW_ZERO is a property of the loaded weights.  nn.Module.load_state_dict on a PARENT (the
    dispatch wrapper of model/__init__.py, as run_v0 --test and train_eval.test do) never calls the child's
    load_state_dict, so the reset must fire from a load_state_dict post-hook."""
    import types
    from pytorch_news_recommender_amd import model as model_pkg
    shape = synth.G1_ODD
    cfg = make_cfg(shape)
    params = synth.make_params(shape, seed=3)
    np.savez(tmp_path / "all_word_embedding_v3.npz", embeddings=params["news_encoder.word_embedding.0.weight"])
    cfg.data_path = str(tmp_path) + "/"
    w = model_pkg.Model(cfg, types.SimpleNamespace(model="nrms_v0", n_GPUs=1))
    w.model._pad_zero = True                       # as established by an earlier forward on a zero padding row
    nz = synth.make_params(shape, seed=4, pad_row_zero=False)
    w.load_state_dict({"model." + k: torch.from_numpy(v) for k, v in nz.items()})
    assert w.model._pad_zero is None               # re-evaluated on the next forward
    w.model._pad_zero = True
    w.model.load_state_dict({k: torch.from_numpy(v) for k, v in nz.items()})
    assert w.model._pad_zero is None


REF = "/root/reference/MIND_2020"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference only exists in the build container")
def test_reference_side_binding_constructs_through_the_reference_wrapper(tmp_path):
    """INTEGRATION.md section 2, executed: a one-line model/nrms_hip.py placed on the REFERENCE's model package
    path, then the reference's own wrapper (model/__init__.py:13-23, imported, not copied) builds it exactly as
    run_v0.py does.  CPU container: construction and checkpoint keys only (forward needs a GPU)."""
    import subprocess
    import sys
    import textwrap
    shape = synth.G1_ODD
    params = synth.make_params(shape, seed=3)
    os.makedirs(tmp_path / "data_processed")
    np.savez(tmp_path / "data_processed" / "all_word_embedding_v3.npz",
             embeddings=params["news_encoder.word_embedding.0.weight"])
    os.makedirs(tmp_path / "binding")
    (tmp_path / "binding" / "nrms_hip.py").write_text(
        "from pytorch_news_recommender_amd.model.nrms_hip import Model  # noqa: F401\n")
    prog = textwrap.dedent("""
        import sys, types
        sys.dont_write_bytecode = True
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import model                                   # the reference's package
        model.__path__.append(%r)                      # where the maintainer's one-line file lives
        from config import Config                      # the reference's config
        cfg = Config('nrms_hip'); cfg.__nrms__()
        cfg.word_embed_size, cfg.num_attention_heads, cfg.query_vector_dim = %d, %d, %d
        args = types.SimpleNamespace(model='NRMS_HIP', n_GPUs=1)
        m = model.Model(cfg, args)
        print('KEYS', ' '.join(sorted(m.state_dict())))
        print('CLASS', type(m.model).__module__)
    """) % (REF, ROOT, str(tmp_path / "binding"), shape.word_embed_size, shape.num_attention_heads,
            shape.query_vector_dim)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", prog], cwd=str(tmp_path), capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    keys = [ln for ln in r.stdout.splitlines() if ln.startswith("KEYS ")][0].split()[1:]
    assert keys == sorted("model." + n for n in synth.param_names())
    assert "CLASS pytorch_news_recommender_amd.model.nrms_hip" in r.stdout


def test_nrms_naml_wrapper_names_layout_and_loud_failure_on_cpu(tmp_path):
    """SURVEY f-3: model.nrms_naml_hip.Model carries the 27 tensors of the reference's nrms_naml.Model.state_dict() (names,
    order, shapes: fixture g7 was loaded into the reference with strict=True under these names), lays them out in one
    flat buffer with the adjacencies the kernels rely on, is reachable through the dispatch wrapper, validates the
    descriptor of its 800-wide user encoder without a GPU, and never computes on the CPU."""
    import types
    from pytorch_news_recommender_amd import model as model_pkg
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.model.nrms_naml_hip import Model
    shape = synth.G7_ODD
    params = synth.make_params_naml(shape, seed=21)
    cfg = Config("nrms_naml")
    cfg.__nrms__()
    assert (cfg.news_feature_size, cfg.query_vector_dim_large, cfg.user_heads_num, cfg.title_heads_num) == (800, 400, 8, 6)
    assert (cfg.category_nums, cfg.subcategory_nums, cfg.cate_embed_size, cfg.n_words_abst) == (19, 294, 100, 40)
    for k in ("word_embed_size", "title_heads_num", "query_vector_dim", "category_nums", "subcategory_nums",
              "cate_embed_size", "user_heads_num", "query_vector_dim_large"):
        setattr(cfg, k, getattr(shape, k))
    cfg.news_feature_size = shape.news_feature_size
    np.savez(tmp_path / "all_word_embedding_v3.npz", embeddings=params["news_encoder.word_embedding.weight"])
    cfg.data_path = str(tmp_path) + "/"
    m = Model(cfg)
    sd = m.state_dict()
    assert list(sd) == list(synth.naml_param_shapes(shape)) and len(sd) == 27
    assert all(tuple(sd[k].shape) == v for k, v in synth.naml_param_shapes(shape).items())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    assert m._views_intact()
    L = m._layout
    for enc in ("news_encoder", "user_encoder"):
        b, w = L.blocks[enc], m._dims.width(enc)
        assert b["wk"] == b["wq"] + w * w and b["wv"] == b["wq"] + 2 * w * w and b["bk"] == b["bq"] + w and b["bv"] == b["bq"] + 2 * w
    assert L.entries["norm.bias"][0] == L.entries["norm.weight"][0] + shape.news_feature_size
    np.testing.assert_array_equal(L.view(m._flat, "norm.weight").numpy(), params["norm.weight"])
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch_naml(shape, seed=22).items()}
    with pytest.raises(_lib.NrmsError, match="no CPU fallback"):
        m(batch)
    cfg.news_feature_size += 4
    with pytest.raises(ValueError, match="news_feature_size"):
        Model(cfg)
    cfg.news_feature_size -= 4
    w = model_pkg.Model(cfg, types.SimpleNamespace(model="nrms_naml", n_GPUs=1))
    assert list(w.state_dict()) == ["model." + k for k in synth.naml_param_shapes(shape)]
    # the library accepts the real widths of the user encoder (d_model 800, d_k 100, q 400) and sizes its workspace
    lib = _lib.load()
    wide = _lib.EncoderDesc(n_seq=512, seq_len=50, d_model=800, n_heads=8, q_dim=400, vocab=0, precision=0,
                            use_output_proj=1, p_drop_attn=0.2, seed=1)
    assert lib.nrms_encoder_bwd_workspace_bytes(ctypes.byref(wide)) >= 4 * 512 * 50 * (800 * 5 + 1)
    bad = _lib.EncoderDesc(n_seq=512, seq_len=50, d_model=800, n_heads=8, q_dim=400, vocab=0, precision=0,
                           use_output_proj=1, p_drop_attn=0.2, p_drop_ctx=0.1, seed=1)
    assert lib.nrms_encoder_bwd_workspace_bytes(ctypes.byref(bad)) == 0 and b"p_drop_ctx" in lib.nrms_last_error()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/nrms_hip.h compiles as C99 (-pedantic) and a C program linked against
    libnrms_hip.so agrees with the ctypes mirror on the descriptor size and gets the argument validation (no GPU needed)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi_check.c"
    src.write_text(
        '#include "nrms_hip.h"\n#include <stdio.h>\n#include <string.h>\n'
        "int main(void) {\n"
        "    nrms_encoder_desc d; memset(&d, 0, sizeof d);\n"
        "    d.n_seq = 4; d.seq_len = 30; d.d_model = 300; d.n_heads = 10; d.q_dim = 200; d.vocab = 100;\n"
        "    size_t ok = nrms_encoder_bwd_workspace_bytes(&d);\n"
        "    d.seq_len = 65;\n"
        "    size_t bad = nrms_encoder_bwd_workspace_bytes(&d);\n"
        '    printf("%zu %zu %zu %s\\n", sizeof(nrms_encoder_desc), ok, bad, nrms_last_error());\n'
        "    return 0;\n}\n")
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "abi_check"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                    str(src), "-o", str(exe), "-L", lib_dir, "-lnrms_hip", "-Wl,-rpath," + lib_dir,
                    "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split(None, 3)
    assert int(out[0]) == ctypes.sizeof(_lib.EncoderDesc) == 72
    assert int(out[1]) > 4 * 120 * 300 * 4 and int(out[2]) == 0 and "seq_len" in out[3]


def test_device_feed_yields_the_batches_of_the_reference_shaped_loader():
    """data_handler.DeviceFeed (news tables and packed samples resident on the device, a batch = row gathers) against
    DataLoader(MyDataset(...)) -- whose items are pinned to the reference's own MyDataset by fixture g6 -- on the same
    samples: identical 13 keys, dtypes and values, training (type 0) and evaluation (type 1) padding, ragged last batch."""
    from torch.utils.data import DataLoader
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.data_handler import DeviceFeed, MyDataset, SyntheticMind
    fx = synth.dataset_fixture_inputs()
    cfg = Config("feed")
    cfg.__nrms__()
    for k, v in fx["config"].items():
        setattr(cfg, k, v)
    for typ, samples in ((0, fx["train_samples"]), (1, fx["eval_samples"])):
        ref = list(DataLoader(MyDataset(cfg, samples, type=typ, id2title_dict=fx["id2title_dict"], id2abst_dict=fx["id2abst_dict"]),
                              batch_size=2, shuffle=False, num_workers=0))
        feed = DeviceFeed(cfg, samples, type=typ, id2title_dict=fx["id2title_dict"], id2abst_dict=fx["id2abst_dict"],
                          batch_size=2, device="cpu")
        got = list(feed)
        assert len(got) == len(ref) == len(feed)
        for a, b in zip(got, ref):
            assert list(a) == list(b)
            for k in b:
                assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), (typ, k)
    # a larger synthetic corpus with categories, shuffled epochs: every sample exactly once per epoch, different orders
    cfg2 = Config("feed2")
    cfg2.__nrms__()
    cfg2.n_words_title, cfg2.n_words = 12, 500
    corpus = SyntheticMind(cfg2, n_news=60, n_topics=3, seed=2, vocab=500)
    samples = corpus.train_samples(37)
    feed = DeviceFeed(cfg2, samples, type=0, id2title_dict=corpus.id2title_dict, id2abst_dict=corpus.id2abst_dict, batch_size=8,
                      device="cpu", shuffle=True, seed=5)
    e1 = torch.cat([b["candidate_ids"] for b in feed])
    e2 = torch.cat([b["candidate_ids"] for b in feed])
    want = torch.tensor([s[3] for s in samples])
    assert e1.shape == want.shape and not torch.equal(e1, e2)
    key = lambda t: sorted(map(tuple, t.tolist()))
    assert key(e1) == key(want) == key(e2)
    b0 = next(iter(DeviceFeed(cfg2, samples, type=0, id2title_dict=corpus.id2title_dict, id2abst_dict=corpus.id2abst_dict,
                              batch_size=8, device="cpu")))
    assert int(b0["browsed_categ_ids"].max()) >= 1 and b0["browsed_absts"].shape == (8, cfg2.history_len, cfg2.n_words_abst)


def test_test_with_pick_best_raises_when_there_is_no_checkpoint(tmp_path):
    """train_eval.test(pick_best=True) with nothing to load must fail loudly (the reference fails on './save_model/' + None,
    train_eval.py:309) instead of scoring whatever weights the model holds and writing a submission file."""
    from pytorch_news_recommender_amd import train_eval
    from pytorch_news_recommender_amd.config import Config
    cfg = Config("nrms_hip")
    cfg.save_path = str(tmp_path / "save_model") + "/"
    with pytest.raises(FileNotFoundError, match="no checkpoint"):
        train_eval.test(cfg, model=None, data_iter=[], pick_best=True)
    os.makedirs(cfg.save_path)
    open(os.path.join(cfg.save_path, "T01-01_00.00_nrms_hip_epoch5_iter_10_auc_0.412.ckpt"), "w").close()   # below 0.5: not a candidate
    with pytest.raises(FileNotFoundError, match="no checkpoint"):
        train_eval.test(cfg, model=None, data_iter=[], pick_best=True)


def test_graph_sampler_draws_co_clicked_news_of_the_batch():
    """graph_sampler.induced_neighbor_rows (the graph encoder's neighbor_rows key, SURVEY f-4 -- no reference counterpart): a
    neighbour is a real click of a user who also clicked the slot's news, never the same news, -1 where nobody clicked it."""
    from pytorch_news_recommender_amd import synth
    from pytorch_news_recommender_amd.graph_sampler import induced_neighbor_rows
    shape = synth.Shape(n_words=12, word_embed_size=8, num_attention_heads=2, query_vector_dim=4, batch_size=40, history_len=9,
                        n_candidates=4, n_words_title=2)
    b = synth.make_batch(shape, seed=3, min_title=1, empty_history_user=True)         # 11 x 11 possible titles: many repeats
    bt, bm, ct = b["browsed_titles"], b["browsed_mask"].astype(bool), b["candidate_titles"]
    B, H, C, K = 40, 9, 4, 6
    nb = induced_neighbor_rows(bt, bm, ct, K, seed=5)
    assert nb.shape == (B * (H + C), K) and nb.dtype == np.int64
    assert np.array_equal(nb, induced_neighbor_rows(bt, bm, ct, K, seed=5))          # a function of the seed
    titles = np.concatenate([bt.reshape(B * H, -1), ct.reshape(B * C, -1)], 0)
    clicked = {}                                                                      # title -> users who clicked it
    for u in range(B):
        for k in range(H):
            if bm[u, k]:
                clicked.setdefault(tuple(bt[u, k]), set()).add(u)
    n_some = 0
    for r in range(B * (H + C)):
        users = clicked.get(tuple(titles[r]), set())
        for v in nb[r]:
            if v < 0:
                continue
            n_some += 1
            assert 0 <= v < B * H and bm.reshape(-1)[v]                               # a real click ...
            assert v // H in users                                                    # ... of a user who clicked this slot's news
            assert tuple(titles[v]) != tuple(titles[r])                               # ... and another news
        if not users:
            assert (nb[r] < 0).all()
    assert n_some > B * H


@pytest.mark.parametrize("name", ["hierec", "graph"])
def test_f4_models_build_on_cpu_and_refuse_to_run_there(name):
    """The row f-4 models (no reference counterpart, parity unpinned) keep the plugin contract -- Model(config), the NRMS news
    encoder's parameter names first, state_dict round trip -- and, like every model of this build, have NO CPU path: a forward
    on CPU parameters raises instead of computing something else."""
    import importlib
    from pytorch_news_recommender_amd import _lib, synth
    from pytorch_news_recommender_amd.config import Config
    shape = synth.Shape(n_words=40, word_embed_size=16, num_attention_heads=2, query_vector_dim=8, batch_size=3, history_len=4,
                        n_candidates=2, n_words_title=5)
    cfg = Config(name)
    cfg.__nrms__()
    cfg.word_embed_size, cfg.num_attention_heads, cfg.query_vector_dim = 16, 2, 8
    cfg.subcategory_nums, cfg.category_nums = 9, 5
    if name == "hierec":
        params = synth.make_params_hierec(shape, 9, 5, seed=1)
        batch = synth.make_batch_hierec(shape, 9, 5, seed=2)
    else:
        params = synth.make_params_graph(shape, seed=1)
        batch = synth.make_batch_graph(shape, 3, seed=2)
    mod = importlib.import_module("pytorch_news_recommender_amd.model." + {"hierec": "hierec_hip", "graph": "graph_hip"}[name])
    m = mod.Model(cfg, pretrained_word_embedding=params["news_encoder.word_embedding.0.weight"])
    res = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    assert not res.missing_keys and not res.unexpected_keys
    names = list(m.state_dict())
    assert names[0] == "news_encoder.word_embedding.0.weight" and set(names) == set(params)
    for k, v in params.items():                                  # parameters are views of one flat buffer, values intact
        assert np.array_equal(m.state_dict()[k].numpy(), v), k
    with pytest.raises(_lib.NrmsError, match="no CPU"):
        m({k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()})
