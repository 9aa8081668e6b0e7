"""Which fp16 roundings make up the score error of nrms_v1's fused news encoder?  The bench-size batch, fp16 mode against the exact
fp32 mode, with subsets of the parameters made exactly fp16-representable beforehand (that operand's rounding then vanishes from
the fp16 kernels while both modes use the same weights).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_v1 import make_v1

shape = synth.Shape(n_words=synth.BENCH.n_words, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                    batch_size=512, history_len=50, n_candidates=5, n_words_title=20)
base = synth.make_params_v1(shape, seed=0)
batch = synth.make_batch(shape, seed=1, mask_some_candidates=True)
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
valid = tb["candidate_mask"] == 1
N = "news_encoder.multi_head_self_attention."
GROUPS = {"table": ["news_encoder.word_embedding.weight"], "W_V": [N + "linear_layers.2.weight"], "W_O": [N + "output_linear.weight"],
          "W_Q,W_K": [N + "linear_layers.0.weight", N + "linear_layers.1.weight"],
          "W_add": ["news_encoder.additive_attention.linear.weight"]}


def run(rounded):
    params = {k: v.copy() for k, v in base.items()}
    for g in rounded:
        for name in GROUPS[g]:
            w = params[name]
            if g == "W_Q,W_K" and name.endswith("0.weight"):       # the kernel rounds W_Q / sqrt(d_k)
                s = np.float32(1.0 / np.sqrt(50.0))
                params[name] = ((w * s).astype(np.float16).astype(np.float32) / s).astype(np.float32)
            else:
                params[name] = w.astype(np.float16).astype(np.float32)
    model = make_v1(shape, params, 6, precision="fp32")
    f = lambda: model.engine.forward(model._flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=False)
    ref = f().clone()
    model.config.precision = "fp16"
    e = (f() - ref)[valid].abs().double()
    return float((e * e).mean().sqrt()), float(e.max()), float((ref[valid].double() ** 2).mean().sqrt())


for rounded in ([], ["table"], ["W_V"], ["W_O"], ["W_V", "W_O"], ["table", "W_V", "W_O"], ["W_Q,W_K"], ["W_add"],
                ["table", "W_V", "W_O", "W_Q,W_K", "W_add"]):
    rms, mx, srms = run(rounded)
    print("exactly representable: %-36s score rms %.3f  error rms %.2e  max %.2e" % (", ".join(rounded) or "(nothing)", srms, rms, mx), flush=True)
