#!/usr/bin/env python3
"""NRMS train-step throughput on MI355X (BASELINE.json metric: users/sec, train step,
MIND shapes hist=50 cand=5 title_len=30 d=300 V=45800, 512 users per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = forward + CE(label 0) + backward + [RCCL all-reduce of the flat gradient] + fused
Adam over all 14.4 M parameters, dropout 0.2 on, synthetic batch already resident in HBM.
Rank 0 prints ONE JSON line; `roofline` is the dominant kernel's measured MFMA rate (HIP events
on the launch stream, over the timed region), `cpu_baseline` is the oracle's reference-shaped
train step timed on this box's host cores (N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from pytorch_news_recommender_amd import parallel, synth
from pytorch_news_recommender_amd.config import Config
from pytorch_news_recommender_amd.model.nrms_hip import Model

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: f32-input MFMA, dense


def kernel_flops(shape, B):
    """Algorithmic flop per launch of each dense-contraction kernel (2mnk)."""
    H, C, L = shape.history_len, shape.n_candidates, shape.n_words_title
    d, q, h = shape.word_embed_size, shape.query_vector_dim, shape.num_attention_heads
    Mn, Mu = B * (H + C) * L, B * H          # news tokens, user-encoder rows
    qkv = lambda M: 2.0 * M * d * 3 * d
    add = lambda M: 2.0 * M * d * q
    att = lambda nseq, S: nseq * 2.0 * 2 * S * S * d     # QK^T + PV over all heads
    return {
        "qkv_proj_fwd": (qkv(Mn), qkv(Mu)),
        "dwqkv_bwd": (qkv(Mn), qkv(Mu)),
        "dx_bwd": (qkv(Mn), qkv(Mu)),
        "addattn_fwd": (add(Mn), add(Mu)),
        "dctx_bwd": (add(Mn), add(Mu)),
        "dwadd_bwd": (add(Mn), add(Mu)),
        "attn_fwd": (att(B * (H + C), L), att(B, H)),
        "attn_bwd": (2.5 * att(B * (H + C), L), 2.5 * att(B, H)),
    }


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU baseline: this process's CPU share (the GPU box gives one GPU's
    slice of the host, far fewer cores than os.cpu_count() reports)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(shape, sample_users=64, steps=3):
    """The oracle's reference-shaped train step (per-slot loop, dropout on, dense per-slot
    embedding grads, torch Adam) on the host cores: bounded sample of the same workload."""
    from oracle import nrms_oracle as orc
    threads = host_threads()
    torch.set_num_threads(threads)
    params = synth.make_params(shape, seed=0)
    trainer = orc.ReferenceShapedTrainer(params, shape.num_attention_heads, p_drop=0.2, lr=1e-3)
    batch = synth.make_batch(shape, seed=1, batch_size=sample_users)
    t0 = time.perf_counter()
    trainer.step(batch)                      # warm-up
    warm = time.perf_counter() - t0
    log("cpu baseline: warm-up step %.1f s on %d threads" % (warm, threads))
    if warm * steps > 40.0:                  # keep the default run within a few minutes
        steps = max(1, int(40.0 / warm))
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step(batch)
    dt = (time.perf_counter() - t0) / steps
    return {"value": sample_users / dt, "unit": "users/s", "cores": threads, "kind": "port",
            "sample": "%d timed train steps of %d users (1 warm-up), same H/C/L/d/V, dropout 0.2, "
                      "per-slot encoder loop + torch Adam, fp32, torch %s CPU" % (steps, sample_users, torch.__version__),
            "ms_per_step": dt * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--users-per-gpu", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-users", type=int, default=64)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16"])
    args = ap.parse_args()

    rank, local_rank, world = parallel.init_process_group(os.environ.get("NRMS_DIST_BACKEND"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev_index = local_rank % max(torch.cuda.device_count(), 1)   # identity on a full node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    shape = synth.BENCH
    B = args.users_per_gpu
    cfg = Config("nrms_hip")
    cfg.__nrms__()
    cfg.n_words_title = shape.n_words_title
    cfg.sample_size = shape.n_candidates - 1
    cfg.batch_size = B
    cfg.dropout = 0.2
    cfg.learning_rate = 1e-3
    cfg.precision = args.precision
    params = synth.make_params(shape, seed=0)
    model = Model(cfg, pretrained_word_embedding=params["news_encoder.word_embedding.0.weight"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to(dev).train()
    model._rank_salt = rank * 0x632BE59BD9B4E019
    eng = model.engine
    parallel.broadcast_parameters(model._flat)
    batch_np = synth.make_batch(shape, seed=1 + rank, batch_size=B)      # each rank: its own users
    batch = {k: torch.from_numpy(v).to(dev) for k, v in batch_np.items()}
    reduce = parallel.GradAllReduce() if world > 1 else None

    def step():
        return model.train_step(batch, world_size=world, all_reduce=reduce)

    log("model ready on %s (rank %d/%d), %d users/GPU" % (dev, rank, world, B))
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    torch.cuda.synchronize()
    log("warm-up done")
    parallel.barrier()
    eng.timing_reset()
    eng.timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_sum = step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    eng.timing(False)
    dt = parallel.max_over_ranks(dt, dev)
    loss = float(loss_sum) / B
    log("timed region: %d steps in %.3f s" % (args.steps, dt))

    if rank == 0:
        flops = kernel_flops(shape, B)
        kernels = {}
        for name, fl in flops.items():
            ms, n = eng.timing_read(name)
            if n == 0:
                continue
            per_step_flop = sum(fl)
            kernels[name] = {"ms_per_step": ms / args.steps, "launches_per_step": n / args.steps,
                             "tflops": per_step_flop / (ms / args.steps * 1e-3) / 1e12}
        for name in ("gather_dropout", "scatter_dropout", "adam", "tn_reduce", "addattn_bwd_rows", "click", "ce_loss",
                     "transpose", "colsum"):
            ms, n = eng.timing_read(name)
            if n:
                kernels[name] = {"ms_per_step": ms / args.steps, "launches_per_step": n / args.steps}
        dom = max((k for k in kernels if "tflops" in kernels[k]), key=lambda k: kernels[k]["ms_per_step"])
        # the dominant kernel's news-encoder launch (the user-encoder launch is ~3% of its flops)
        dom_ms, dom_n = eng.timing_read(dom)
        achieved = sum(flops[dom]) * args.steps / (dom_ms * 1e-3) / 1e12
        total_users = B * world * args.steps
        out = {
            "metric": "users/sec (train step) NRMS MIND-small hist=50 cand=5",
            "value": total_users / dt, "unit": "users/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1] shapes at fp32: NRMS(nrms_v0) train step, %d users/GPU, hist=50, "
                                   "cand=5, title_len=30, d=300, h=10, q=200, V=45800, dropout=0.2, Adam(lr=1e-3)" % B,
                       "users_per_gpu": B, "global_batch": B * world,
                       "parallelism": "dp%d" % world, "precision": "fp32 MFMA (exact f32)"},
            "loss": loss,
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "avg_launch_ms": dom_ms / max(dom_n, 1),
                         "algorithmic_flop_per_step": sum(flops[dom])},
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shape, sample_users=args.cpu_sample_users)
        print(json.dumps(out), flush=True)
    parallel.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
