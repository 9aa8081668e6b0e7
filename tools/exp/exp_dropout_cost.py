"""How much of the fused kernels' time is the context dropout (Philox)?  The bench step with dropout 0.2 and 0.0, per-kernel
timers, helper streams off.  GPU box only."""
import os, sys
os.environ["NRMS_NO_SIDE_STREAMS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

shape = synth.BENCH
params = synth.make_params(shape, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in synth.make_batch(shape, seed=1, batch_size=512).items()}
for p in (0.2, 0.0):
    m = make_model(shape, params, dropout=p, precision="fp16").train()
    for _ in range(3):
        m.train_step(batch)
    eng = m.engine
    eng.timing(True); eng.timing_reset()
    for _ in range(5):
        m.train_step(batch)
    torch.cuda.synchronize()
    print("dropout %.1f: " % p + "  ".join("%s %.3f" % (n, eng.timing_read(n)[0] / 5) for n in
          ("fused_fwd16", "fused_bwd16_pool", "fused_bwd16_attn", "gather_dropout", "scatter_dropout")))
    eng.timing(False)
