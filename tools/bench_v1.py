"""Times nrms_v1's Model.train_step (bench.py's v1 variant: 20-word titles, 6 title heads of 50 / 10 user heads of 30, W_O,
masks, dropout 0.2) with the per-kernel timers.  GPU box only.  Usage: python tools/bench_v1.py [fp16|fp32|bf16x3] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_news_recommender_amd import synth
from tests.test_hip_v1 import make_v1

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
shape = synth.Shape(n_words=synth.BENCH.n_words, word_embed_size=300, num_attention_heads=10, query_vector_dim=200,
                    batch_size=B, history_len=50, n_candidates=5, n_words_title=20)
params = synth.make_params_v1(shape, seed=0)
batch = {k: torch.from_numpy(v).cuda() for k, v in synth.make_batch(shape, seed=1, batch_size=B).items()}
model = make_v1(shape, params, 6, dropout=0.2, precision=prec).train()
for _ in range(3):
    model.train_step(batch)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 10
ev[0].record()
for _ in range(n):
    ls = model.train_step(batch)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / n
print("nrms_v1 %-7s train step: %.2f ms -> %.0f users/s   loss %.4f" % (prec, ms, B / ms * 1e3, float(ls) / B), flush=True)
eng = model.engine
eng.timing(True)
eng.timing_reset()
for _ in range(3):
    model.train_step(batch)
torch.cuda.synchronize()
rows = []
for name in ("qkv_proj_fwd", "out_proj_fwd", "attn_fwd", "attn_bwd", "addattn_fwd", "addattn_bwd_rows", "dctx_bwd", "dwadd_bwd",
             "dwo_bwd", "dattn_bwd", "dwqkv_bwd", "dx_bwd", "gather_dropout", "scatter_dropout", "fill_pad_rows", "transpose",
             "permute_rows", "split_planes", "compact_rows", "sanitize_ids", "click", "ce_loss", "adam", "tn_reduce", "padsum_reduce",
             "colsum", "colsum_add", "fused_fwd16", "fused_bwd16_pool", "fused_bwd16_attn", "prep16", "red16", "closed16"):
    t, k = eng.timing_read(name)
    if k:
        rows.append((t / 3, name, k / 3))
for t, name, k in sorted(rows, reverse=True):
    print("   %-18s %8.3f ms/step  (%.0f launches)" % (t, name, k) if False else "   %-18s %8.3f ms/step  (%.0f launches)" % (name, t, k))
print("   sum of timed kernels: %.2f ms" % sum(r[0] for r in rows))
