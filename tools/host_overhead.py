"""Host time to ENQUEUE one fp16 train step (Python + ctypes + launches) against the GPU time of the step: the margin by
which the host stays ahead of the device.  GPU box only.  Usage: python tools/host_overhead.py [precision]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
shape = synth.BENCH
params = synth.make_params(shape, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in synth.make_batch(shape, seed=1).items()}
model = make_model(shape, params, dropout=0.2, precision=prec).train()
for _ in range(5):
    model.train_step(batch)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    model.train_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s: host enqueue %.3f ms/step, wall %.3f ms/step (GPU-bound if enqueue < wall)" % (prec, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
# host-only cost: enqueue while the GPU is kept idle-free is not separable, so also time a burst of steps after a sync with
# the profiler's view: cProfile of 20 steps
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    model.train_step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(18)
