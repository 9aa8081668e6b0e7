"""Score error of the fp16 mode against the library's exact fp32 mode at the bench shape, three seeds, with the user encoder
in bf16x3 (the default) and in fp16 (config.fp16_user_encoder).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model
shape = synth.BENCH
for seed in (0, 7, 13):
    params = synth.make_params(shape, seed=seed)
    batch = synth.make_batch(shape, seed=1 + seed, mask_some_candidates=True)
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    model = make_model(shape, params).eval()
    out = {}
    for prec, u16 in (("fp32", False), ("fp16", False), ("fp16", True), ("bf16x3", False)):
        model.config.precision = prec
        model.config.fp16_user_encoder = u16
        with torch.no_grad():
            model.dedup_inference = False
            out[(prec, u16)] = model(tb).double().cpu().numpy()
    valid = batch["candidate_mask"] == 1
    ref = out[("fp32", False)]
    for key in (("fp16", False), ("fp16", True), ("bf16x3", False)):
        e = np.abs(out[key] - ref)[valid]
        print("seed %2d %-7s user16=%-5s score rms %.3f  max|err| %.3e  p99.9 %.3e  rms err %.3e  n=%d  frac>7e-5: %.4f" % (
            seed, key[0], key[1], np.sqrt((ref[valid] ** 2).mean()), e.max(), np.quantile(e, 0.999), np.sqrt((e ** 2).mean()), e.size, (e > 7e-5).mean()))
