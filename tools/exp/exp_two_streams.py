"""Would micro-batch pipelining pay?  Two independent models, 256 users each, stepping concurrently on two HIP streams (one Python
thread each) against one model stepping 512 users -- an upper bound of what overlapping one half-batch's user-encoder phases with
the other's news-encoder phases could give.  GPU box only."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

shape = synth.BENCH


def build(B, seed):
    m = make_model(shape, synth.make_params(shape, seed=seed), dropout=0.2, precision="fp16").train()
    batch = {k: torch.from_numpy(v).cuda() for k, v in synth.make_batch(shape, seed=seed + 1, batch_size=B).items()}
    return m, batch


def run_single(B, n):
    m, batch = build(B, 0)
    for _ in range(5):
        m.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.train_step(batch)
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


def run_pair(B, n):
    ms = [build(B, 0), build(B, 10)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for (m, batch), s in zip(ms, streams):
        with torch.cuda.stream(s):
            for _ in range(5):
                m.train_step(batch)
    torch.cuda.synchronize()

    def work(i):
        m, batch = ms[i]
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                m.train_step(batch)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return 2 * B * n / (time.perf_counter() - t0)


print("one model, 512 users/step:            %.0f users/s" % run_single(512, 40))
print("one model, 256 users/step:            %.0f users/s" % run_single(256, 40))
print("two models x 256 users, two streams:  %.0f users/s" % run_pair(256, 40))
print("two models x 512 users, two streams:  %.0f users/s" % run_pair(512, 40))
