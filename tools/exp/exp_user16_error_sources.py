"""Which fp16 roundings of the USER encoder make up what it adds to the score error when it runs on the fused fp16 kernels
(config.fp16_user_encoder)?  Bench-size batch, three seeds, fp16 mode against the exact fp32 mode on the same weights, with
subsets of the user encoder's parameters made exactly fp16-representable beforehand (that operand's rounding then vanishes).
GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

shape = synth.BENCH
U = "user_encoder.multihead_self_attention."
GROUPS = {"uW_V": [U + "W_V.weight"], "uW_Q,uW_K": [U + "W_Q.weight", U + "W_K.weight"],
          "uW_add": ["user_encoder.additive_attention.linear.weight"]}


def run(seed, rounded, user16):
    base = synth.make_params(shape, seed=seed)
    batch = synth.make_batch(shape, seed=1 + seed, mask_some_candidates=True)
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    valid = tb["candidate_mask"] == 1
    params = {k: v.copy() for k, v in base.items()}
    for g in rounded:
        for name in GROUPS[g]:
            w = params[name]
            if name.endswith("W_Q.weight"):
                s = np.float32(1.0 / np.sqrt(30.0))
                params[name] = ((w * s).astype(np.float16).astype(np.float32) / s).astype(np.float32)
            else:
                params[name] = w.astype(np.float16).astype(np.float32)
    model = make_model(shape, params, precision="fp32", fp16_user=user16)
    f = lambda: model.engine.forward(model._flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=True)
    ref = f().clone()
    model.config.precision = "fp16"
    e = (f() - ref)[valid].abs().double()
    return float((e * e).mean().sqrt()), float(e.max())


for user16, rounded in ((False, []), (True, []), (True, ["uW_V"]), (True, ["uW_Q,uW_K"]), (True, ["uW_add"]),
                        (True, ["uW_V", "uW_Q,uW_K", "uW_add"])):
    res = [run(s, rounded, user16) for s in (0, 7, 13)]
    print("user encoder %-6s exactly representable: %-28s error rms %s  max %s" % (
        "fp16" if user16 else "bf16x3", ", ".join(rounded) or "(nothing)", " ".join("%.2e" % r[0] for r in res),
        " ".join("%.2e" % r[1] for r in res)), flush=True)
