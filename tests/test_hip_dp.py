"""Data-parallel HIP training: 2 ranks (both on the box's single GPU, gloo exchange staged through
the host) must land on the same parameters as 1 process stepping the global batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(shape, params, precision="fp32"):
    from pytorch_news_recommender_amd.config import Config
    from pytorch_news_recommender_amd.model.nrms_hip import Model
    cfg = Config("nrms_hip")
    cfg.__nrms__()
    cfg.word_embed_size, cfg.num_attention_heads, cfg.query_vector_dim = (
        shape.word_embed_size, shape.num_attention_heads, shape.query_vector_dim)
    cfg.dropout = 0.0
    cfg.precision = precision
    m = Model(cfg, pretrained_word_embedding=params["news_encoder.word_embedding.0.weight"])
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return m.to("cuda:0").train()


def _sync(kind, model, force=False):
    """"allreduce": parallel.GradAllReduce (+ full Adam on every rank); "sharded": reduce-scatter, Adam on the owned
    1/world, all-gather (parallel.ShardedGradSync)."""
    from pytorch_news_recommender_amd import parallel
    if kind == "allreduce":
        return parallel.GradAllReduce(force=force)
    n_table = model._dims.n_words * model._dims.word_embed_size
    return parallel.ShardedGradSync(model._flat.numel(), n_table, force=force)


def _worker(rank, world, port, out_dir, precision, sync="allreduce"):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from pytorch_news_recommender_amd import parallel, synth
    parallel.init_process_group(backend="gloo")
    shape = synth.Shape(n_words=400, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=6, history_len=50, n_candidates=5, n_words_title=30)
    params = synth.make_params(shape, seed=9 + rank)          # deliberately different: broadcast must fix it
    model = _build(shape, params, precision)
    model.engine
    parallel.broadcast_parameters(model._flat, src=0)
    reduce = _sync(sync, model)
    losses = []
    for t in range(2):
        gbatch = synth.make_batch(shape, seed=20 + t, ragged=True)
        local = {k: torch.from_numpy(v) for k, v in parallel.shard_batch(gbatch, rank, world).items()}
        ls = model.train_step(local, world_size=world, all_reduce=reduce, global_batch=shape.batch_size)
        losses.append(float(ls))
    flat = model._flat.detach().cpu().numpy()
    np.save(os.path.join(out_dir, "flat%d.npy" % rank), flat)
    np.save(os.path.join(out_dir, "loss%d.npy" % rank), np.array(losses))
    parallel.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("precision,sync", [("fp32", "allreduce"), ("fp16", "allreduce"), ("fp32", "sharded"), ("fp16", "sharded")])
def test_two_ranks_match_single_process(tmp_path, precision, sync):
    from pytorch_news_recommender_amd import synth
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), precision, sync), nprocs=2, join=True)
    f0, f1 = np.load(tmp_path / "flat0.npy"), np.load(tmp_path / "flat1.npy")
    assert np.abs(f0 - f1).max() < 1e-7                   # replicas stay in lock-step
    shape = synth.Shape(n_words=400, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=6, history_len=50, n_candidates=5, n_words_title=30)
    model = _build(shape, synth.make_params(shape, seed=9), precision)
    tot = []
    for t in range(2):
        gbatch = {k: torch.from_numpy(v) for k, v in synth.make_batch(shape, seed=20 + t, ragged=True).items()}
        tot.append(float(model.train_step(gbatch)))
    single = model._flat.detach().cpu().numpy()
    l0, l1 = np.load(tmp_path / "loss0.npy"), np.load(tmp_path / "loss1.npy")
    np.testing.assert_allclose(l0 + l1, tot, rtol=1e-5 if precision == "fp32" else 1e-4)   # local loss sums add up to the global sum
    diff = np.abs(f0 - single)
    # Adam amplifies summation-order noise where |g| ~ eps (see test_hip_parity.assert_params_close); in fp16 mode the two
    # runs also round their gradients differently (the loss scale follows the local batch): sign flips of noise-level
    # gradient elements move a parameter by up to lr per step
    if precision == "fp32":
        assert np.median(diff) < 1e-6 and np.quantile(diff, 0.999) < 1e-4 and diff.max() < 2.1e-3, (
            float(np.median(diff)), float(diff.max()))
    else:
        assert np.median(diff) < 2e-5 and diff.max() < 4.1e-3, (float(np.median(diff)), float(diff.max()))


def test_deferred_wqkv_backward_equals_plain_backward():
    """engine.backward(table_grad_ready=...) -- NRMS_FLAG_DEFER_WQKV + nrms_encoder_bwd_wqkv -- calls the hook
    once, after which only d(W_qkv) / d(b_qkv) of the news encoder are still outstanding; the result equals the
    one-call backward (bit for bit outside the float-atomic table gradient)."""
    import numpy as np
    import torch
    from pytorch_news_recommender_amd import synth
    from tests.test_hip_parity import make_model, tbatch
    shape = synth.Shape(n_words=400, word_embed_size=120, num_attention_heads=6, query_vector_dim=64,
                        batch_size=9, history_len=12, n_candidates=4, n_words_title=14)
    params = synth.make_params(shape, seed=141)
    batch = tbatch(synth.make_batch(shape, seed=142, ragged=True, min_title=1))
    model = make_model(shape, params)
    eng, flat, lay = model.engine, model._flat, model._layout
    dev = flat.device
    bt, ct, cm = (batch[k].to(dev) for k in ("browsed_titles", "candidate_titles", "candidate_mask"))
    s = eng.forward(flat, bt, ct, cm, training=True)
    dsc = (torch.randn(s.shape, generator=torch.Generator().manual_seed(3)) * 1e-2).to(dev)
    g_plain = torch.zeros_like(flat)
    eng.backward(flat, g_plain, dsc)
    eng.forward(flat, bt, ct, cm, training=True)
    g_split = torch.zeros_like(flat)
    seen = []

    def hook():
        torch.cuda.synchronize()
        wq = lay.view(g_split, "news_encoder.multihead_self_attention.W_Q.weight")
        tab = lay.view(g_split, "news_encoder.word_embedding.0.weight")
        seen.append((float(wq.abs().max()), float(tab.abs().max())))

    eng.backward(flat, g_split, dsc, table_grad_ready=hook)
    assert len(seen) == 1 and seen[0][0] == 0.0 and seen[0][1] > 0.0      # table done, W_qkv still outstanding
    table = "news_encoder.word_embedding.0.weight"
    for n in lay.names:
        a, b = lay.view(g_split, n), lay.view(g_plain, n)
        if n == table:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-9)
        else:
            assert torch.equal(a, b), n


def _rccl_worker(rank, port, out_dir, precision, sync="allreduce"):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pytorch_news_recommender_amd import parallel, synth
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    shape = synth.Shape(n_words=400, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=6, history_len=50, n_candidates=5, n_words_title=30)
    model = _build(shape, synth.make_params(shape, seed=9), precision)
    model.engine
    parallel.broadcast_parameters(model._flat, src=0)
    reduce = _sync(sync, model, force=True)
    assert reduce.active
    losses = []
    for t in range(3):
        gbatch = {k: torch.from_numpy(v) for k, v in synth.make_batch(shape, seed=20 + t, ragged=True).items()}
        losses.append(float(model.train_step(gbatch, world_size=1, all_reduce=reduce)))
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "rccl_flat.npy"), model._flat.detach().cpu().numpy())
    np.save(os.path.join(out_dir, "rccl_loss.npy"), np.array(losses))
    dist.destroy_process_group()


@pytest.mark.parametrize("precision,sync", [("fp32", "allreduce"), ("fp16", "allreduce"), ("fp16", "sharded")])
def test_rccl_collectives_on_one_gpu(tmp_path, precision, sync):
    """The data-parallel step through REAL RCCL calls (backend "nccl", a one-rank group: two ranks cannot share a GPU under
    RCCL): asynchronous all-reduce of the table gradient started from inside the backward, the deferred d(W_qkv) GEMMs
    enqueued under it, the second all-reduce, the waits, Adam -- must equal the plain single-process step
    (a one-rank sum is the identity), i.e. the stream ordering between the kernels and RCCL's stream holds.
    sync = "sharded": the same through ncclReduceScatter / ncclAllGather (parallel.ShardedGradSync; one rank owns it all)."""
    from pytorch_news_recommender_amd import synth
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path), precision, sync), nprocs=1, join=True)
    shape = synth.Shape(n_words=400, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                        batch_size=6, history_len=50, n_candidates=5, n_words_title=30)
    model = _build(shape, synth.make_params(shape, seed=9), precision)
    tot = []
    for t in range(3):
        gbatch = {k: torch.from_numpy(v) for k, v in synth.make_batch(shape, seed=20 + t, ragged=True).items()}
        tot.append(float(model.train_step(gbatch)))
    single = model._flat.detach().cpu().numpy()
    got = np.load(tmp_path / "rccl_flat.npy")
    np.testing.assert_allclose(np.load(tmp_path / "rccl_loss.npy"), tot, rtol=1e-6)
    diff = np.abs(got - single)
    print("rccl one-rank vs plain: median %.3g  p99.9 %.3g  max %.3g" % (np.median(diff), np.quantile(diff, 0.999), diff.max()))
    assert diff.max() < 1e-7, float(diff.max())           # measured: 0 (fp32), 2.3e-10 (fp16)
