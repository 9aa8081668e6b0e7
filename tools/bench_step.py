"""Times Model.train_step at the bench shape per precision mode, with the per-kernel timers.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

shape = synth.BENCH
params = synth.make_params(shape, seed=0)
batch = synth.make_batch(shape, seed=1)
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
MODES = sys.argv[1:] or ["bf16x3", "fp16"]
for prec in MODES:
    model = make_model(shape, params, dropout=0.2, precision=prec).train()
    for _ in range(3):
        model.train_step(tb)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 10
    ev[0].record()
    for _ in range(n):
        ls = model.train_step(tb)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    print("%-7s train step: %.3f ms  -> %.0f users/s   loss %.4f" % (prec, ms, 512 / ms * 1e3, float(ls) / 512), flush=True)
    eng = model.engine
    eng.timing(True); eng.timing_reset()
    for _ in range(5):
        model.train_step(tb)
    torch.cuda.synchronize()
    rows = []
    for name in ("fused64_fwd16", "fused64_bwd16_pool", "fused64_bwd16_attn", "fused_fwd16", "fused_bwd16_pool", "fused_bwd16_attn", "dwqkv_bwd", "dwadd_bwd", "dx_bwd", "tn_reduce", "red16", "prep16", "gather_dropout",
                 "scatter_dropout", "compact_rows", "title_order", "sanitize_ids", "cast16", "adam", "qkv_proj_fwd", "attn_fwd",
                 "attn_bwd", "addattn_fwd", "addattn_bwd_rows", "dctx_bwd", "fill_pad_rows", "click", "ce_loss", "split_planes",
                 "transpose", "permute_rows", "colsum", "padsum"):
        t, k = eng.timing_read(name)
        if k:
            rows.append((t / 5, name, k / 5))
    tot = sum(r[0] for r in rows)
    for t, name, k in sorted(rows, reverse=True):
        print("   %-18s %.3f ms/step  (%.0f launches)" % (name, t, k))
    print("   sum of timed kernels: %.3f ms" % tot)
    eng.timing(False)
