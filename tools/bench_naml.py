"""Times nrms_naml's Model.train_step (SURVEY f-3) at the reference's shapes (B=512, H=50, C=5, title 20, abstract 40,
d=300, 800-wide user encoder) with the per-kernel timers.  GPU box only.  Usage: python tools/bench_naml.py [fp32|bf16x3] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_news_recommender_amd import synth
from tests.test_hip_naml import make_model

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
shape = synth.NamlShape(batch_size=B)
params = synth.make_params_naml(shape, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in synth.make_batch_naml(shape, seed=1).items()}
model = make_model(shape, params, dropout=0.2, precision=prec).train()
for _ in range(2):
    model.train_step(batch)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 5
ev[0].record()
for _ in range(n):
    ls = model.train_step(batch)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / n
print("nrms_naml %-7s train step: %.2f ms -> %.0f users/s   loss %.4f" % (prec, ms, B / ms * 1e3, float(ls) / B), flush=True)
eng = model.engine
eng.timing(True)
eng.timing_reset()
for _ in range(3):
    model.train_step(batch)
torch.cuda.synchronize()
rows = []
for name in ("qkv_proj_fwd", "out_proj_fwd", "attn_fwd", "attn_bwd", "wide_attn_fwd", "wide_attn_bwd", "addattn_fwd", "addattn_proj_fwd", "addattn_bwd_rows",
             "dctx_bwd", "dwadd_bwd", "dwo_bwd", "dattn_bwd", "dwqkv_bwd", "dx_bwd", "gather_dropout", "scatter_dropout",
             "features_fwd", "features_bwd", "layernorm_fwd", "layernorm_bwd", "colsum_add", "transpose", "permute_rows",
             "split_planes", "empty_seq_fwd", "empty_seq_bwd", "title_order", "compact_rows", "sanitize_ids", "click", "ce_loss", "adam", "tn_reduce"):
    t, k = eng.timing_read(name)
    if k:
        rows.append((t / 3, name, k / 3))
for t, name, k in sorted(rows, reverse=True):
    print("   %-18s %8.3f ms/step  (%.0f launches)" % (name, t, k))
print("   sum of timed kernels: %.2f ms" % sum(r[0] for r in rows))
