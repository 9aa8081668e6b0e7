import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model
shape = synth.BENCH
for seed in (0, 7):
    params = synth.make_params(shape, seed=seed)
    batch = synth.make_batch(shape, seed=1 + seed, mask_some_candidates=True)
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    model = make_model(shape, params).eval()
    out = {}
    for prec in ("fp32", "fp16", "bf16x3"):
        model.config.precision = prec
        with torch.no_grad():
            model.dedup_inference = False
            out[prec] = model(tb).double().cpu().numpy()
    valid = batch["candidate_mask"] == 1
    for prec in ("fp16", "bf16x3"):
        e = np.abs(out[prec] - out["fp32"])[valid]
        print("seed %d %s: score rms %.3f  max|err| %.3e  p99.9 %.3e  rms err %.3e  n=%d  frac>1e-4: %.4f" % (
            seed, prec, np.sqrt((out["fp32"][valid] ** 2).mean()), e.max(), np.quantile(e, 0.999), np.sqrt((e ** 2).mean()), e.size, (e > 1e-4).mean()))
