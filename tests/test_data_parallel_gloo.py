"""world_size-2 data-parallel plumbing on CPU (gloo): user sharding, the single flat-gradient
all-reduce, parameter broadcast and max-over-ranks timing of pytorch_news_recommender_amd.parallel.
The gradient provider here is the oracle (no GPU in this container); the identity under test is
the one the HIP path relies on:  sum_r grad_r(shard_r, scale 1/B_global) == grad(global batch)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import nrms_oracle as orc
    from pytorch_news_recommender_amd import parallel, synth
    from pytorch_news_recommender_amd.engine import FlatLayout, ModelDims
    r, lr, w = parallel.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    shape = synth.Shape(n_words=120, word_embed_size=60, num_attention_heads=6, query_vector_dim=32,
                        batch_size=5, history_len=6, n_candidates=3, n_words_title=7)   # 5 users: uneven shards
    lay = FlatLayout(ModelDims(shape.n_words, shape.word_embed_size, shape.num_attention_heads,
                               shape.query_vector_dim))
    # replicas start from rank 0's parameters
    params = synth.make_params(shape, seed=100 + rank)
    flat = torch.zeros(lay.total)
    for n, v in params.items():
        lay.view(flat, n).copy_(torch.from_numpy(v))
    parallel.broadcast_parameters(flat, src=0)
    params = {n: lay.view(flat, n).numpy().copy() for n in params}
    ref0 = synth.make_params(shape, seed=100)
    assert all(np.array_equal(params[n], ref0[n]) for n in params)

    gbatch = synth.make_batch(shape, seed=7, ragged=True, min_title=1, mask_some_candidates=True)
    local = parallel.shard_batch(gbatch, rank, world)
    lo, hi = parallel.shard_rows(shape.batch_size, rank, world)
    assert len(local["browsed_titles"]) == hi - lo
    # local gradient of (sum of local losses) / B_global
    p = orc.to_torch(params, requires_grad=True)
    scores, _ = orc.forward(p, local, shape.num_attention_heads)
    loss_sum = torch.nn.functional.cross_entropy(scores, torch.zeros(len(scores), dtype=torch.long), reduction="sum")
    (loss_sum / shape.batch_size).backward()
    gflat = torch.zeros(lay.total)
    for n, t in p.items():
        if t.grad is not None:
            lay.view(gflat, n).copy_(t.grad)
    # the way Model.train_step reduces: table part asynchronously (overlapped with the deferred d(W_qkv)
    # GEMM on the GPU), the remaining tensors afterwards, then wait
    reduce = parallel.GradAllReduce()
    n_table = shape.n_words * shape.word_embed_size
    assert lay.entries["news_encoder.word_embedding.0.weight"][0] == 0          # the table leads the flat buffer
    handle = reduce.start(gflat[:n_table])
    reduce(gflat[n_table:])
    handle.wait()
    tmax = parallel.max_over_ranks(float(rank + 1), torch.device("cpu"))
    assert tmax == float(world)
    parallel.barrier()
    if rank == 0:
        _, _, grads, _ = orc.loss_and_grads(params, gbatch, shape.num_attention_heads)
        for n in params:
            np.testing.assert_allclose(lay.view(gflat, n).numpy(), grads[n], rtol=1e-4, atol=2e-7, err_msg=n)
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.destroy_process_group()


def test_two_rank_gradient_equals_global_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_rows_cover_and_balance():
    sys.path.insert(0, ROOT)
    from pytorch_news_recommender_amd import parallel
    for n in (0, 1, 5, 512, 4096, 4099):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
