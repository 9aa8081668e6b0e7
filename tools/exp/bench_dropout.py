"""A/B: the fp16 train step with and without dropout (how much of the step is the counter-based RNG).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

shape = synth.BENCH
params = synth.make_params(shape, seed=0)
tb = {k: torch.from_numpy(v).cuda() for k, v in synth.make_batch(shape, seed=1).items()}
for p in (0.2, 0.0):
    model = make_model(shape, params, dropout=p, precision="fp16").train()
    for _ in range(3):
        model.train_step(tb)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        model.train_step(tb)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    eng = model.engine
    os.environ["NRMS_NO_SIDE_STREAMS"] = "1"
    eng.timing(True); eng.timing_reset()
    for _ in range(5):
        model.train_step(tb)
    torch.cuda.synchronize()
    eng.timing(False)
    os.environ.pop("NRMS_NO_SIDE_STREAMS")
    t = {n: eng.timing_read(n)[0] / 5 for n in ("fused_fwd16", "fused_bwd16_pool", "fused_bwd16_attn", "gather_dropout", "scatter_dropout")}
    print("dropout %.1f: %.3f ms/step  " % (p, ms) + "  ".join("%s %.3f" % kv for kv in t.items()), flush=True)
