"""Times the news-encoder forward (all titles of a bench-shaped batch) per precision mode.  GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

shape = synth.BENCH
params = synth.make_params(shape, seed=0)
batch = synth.make_batch(shape, seed=1)
ids = torch.from_numpy(np.concatenate([batch["browsed_titles"].reshape(-1, 30), batch["candidate_titles"].reshape(-1, 30)])).cuda()
ref = None
MODES = sys.argv[1:] or ["fp32", "bf16x3", "fp16"]
for prec in MODES:
    model = make_model(shape, params, precision=prec).eval()
    eng = model.engine
    out = eng.encode_titles(model._flat, ids, chunk_titles=1 << 20)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    err = float((out - ref).abs().max()) if prec != MODES[0] else 0.0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        eng.encode_titles(model._flat, ids, chunk_titles=1 << 20)
    ev[0].record()
    n = 10
    for _ in range(n):
        eng.encode_titles(model._flat, ids, chunk_titles=1 << 20)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    print("%-7s news-encoder forward, %d titles: %.3f ms  (max |vec - fp32| = %.2e)" % (prec, ids.shape[0], ms, err), flush=True)
    if prec == "fp16":
        eng.timing(True); eng.timing_reset()
        for _ in range(5):
            eng.encode_titles(model._flat, ids, chunk_titles=1 << 20)
        for name in ("fused_fwd16", "gather_dropout", "prep16", "compact_rows", "sanitize_ids"):
            t, k = eng.timing_read(name)
            print("   %-16s %.3f ms/launch x %d" % (name, t / max(k, 1), k))
        eng.timing(False)
