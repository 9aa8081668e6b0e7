"""Fused split-bf16 user encoder (csrc/user64.hip, NRMS_FLAG_FUSED_SEQ64) against the unfused bf16x3 chain and the exact fp32
mode: user vectors, d(input), all user-encoder gradients; then kernel times.  GPU box only.
    python tools/debug_user64.py [B] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 50
shape = synth.Shape(n_words=1000, word_embed_size=300, num_attention_heads=10, query_vector_dim=200, batch_size=B,
                    history_len=H, n_candidates=5, n_words_title=30)
params = synth.make_params(shape, seed=3)
model = make_model(shape, params, precision="bf16x3")
eng, flat, lay = model.engine, model._flat, model._layout
g = torch.Generator().manual_seed(5)
x = (torch.randn(B, H, 300, generator=g) * 0.3).cuda()
dout = (torch.randn(B, 300, generator=g) * 1e-3).cuda()


def run(fused, prec="bf16x3"):
    model.config.precision = prec
    e = model.engine
    e.fused_user_encoder = fused
    out = e.encode_users(flat, x, save=True, tag="dbg").clone()
    gf = torch.zeros_like(flat)
    dx = e.encode_users_backward(flat, gf, x, dout, tag="dbg").clone()
    torch.cuda.synchronize()
    return out, dx, gf


ref = run(False, "fp32")
chain = run(False)
fused = run(True)
assert eng._desc("user_encoder", B, H, training=True).flags & 8
for name, a, b in (("chain vs fp32", chain, ref), ("fused vs fp32", fused, ref), ("fused vs chain", fused, chain)):
    eo = float((a[0] - b[0]).abs().max()); so = float(b[0].abs().max())
    ex = float((a[1] - b[1]).abs().max()); sx = float(b[1].abs().max())
    print("%-15s out max err %.2e (scale %.2e)  dx max err %.2e (scale %.2e)" % (name, eo, so, ex, sx))
    for n in lay.names:
        if not n.startswith("user_encoder."):
            continue
        ga, gb = lay.view(a[2], n), lay.view(b[2], n)
        print("      %-60s err %.2e of scale %.2e" % (n, float((ga - gb).abs().max()), float(gb.abs().max())))
# inference path (no saved activations)
model.config.precision = "bf16x3"
e = model.engine
e.fused_user_encoder = True
oi = e.encode_users(flat, x)
print("inference vs training forward: max diff %.2e" % float((oi - fused[0]).abs().max()))
# determinism
f2 = run(True)
print("bit-reproducible:", all(torch.equal(p, q) for p, q in zip(fused, f2)))
# timing
for fz in (False, True):
    e.fused_user_encoder = fz
    for _ in range(3):
        run(fz)
    e.timing_reset(); e.timing(True)
    n = 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = e.encode_users(flat, x, save=True, tag="dbg")
        gf = torch.zeros_like(flat)
        e.encode_users_backward(flat, gf, x, dout, tag="dbg")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    e.timing(False)
    names = ["user64_prep", "user64_fwd", "user64_bwd", "qkv_proj_fwd", "attn_fwd", "addattn_fwd", "addattn_bwd_rows", "dctx_bwd", "attn_bwd",
             "dwadd_bwd", "dwqkv_bwd", "dx_bwd", "transpose", "permute_rows", "split_planes", "colsum", "tn_reduce"]
    parts = []
    for nm in names:
        ms, cnt = e.timing_read(nm)
        if cnt:
            parts.append("%s %.3f" % (nm, ms / n))
    print("fused=%s: fwd+bwd %.3f ms wall (with timers)  | %s" % (fz, dt * 1e3, "  ".join(parts)))
