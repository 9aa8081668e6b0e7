"""fp16-mode gradients against the fp32 mode at the bench shape, tensor by tensor (relative rms error), for a CE-shaped
upstream gradient and a range of loss scales.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

shape = synth.BENCH
params = synth.make_params(shape, seed=0)
batch = synth.make_batch(shape, seed=1)
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
model = make_model(shape, params)
lay = model._layout


def grads(prec, scale=None):
    model.config.precision = prec
    eng = model.engine
    if scale is not None:
        eng.loss_scale_override = scale
    s = eng.forward(model._flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=True)
    _, ds = eng.ce_loss(s, grad_scale=1.0 / shape.batch_size)
    g = torch.zeros_like(model._flat)
    eng.backward(model._flat, g, ds)
    return g

ref = grads("fp32")
for scale in (2.0 ** 16, 2.0 ** 20, 2.0 ** 24, 2.0 ** 28):
    g = grads("fp16", scale)
    row = []
    for n in lay.names:
        a, b = lay.view(g, n).double(), lay.view(ref, n).double()
        rel = float(((a - b) ** 2).mean().sqrt() / ((b ** 2).mean().sqrt() + 1e-300))
        row.append((n.replace("multihead_self_attention.", "").replace("additive_attention.", "add."), float((b ** 2).mean().sqrt()), rel))
    print("loss scale 2^%d  finite=%s" % (int(torch.log2(torch.tensor(scale))), bool(torch.isfinite(g).all())))
    print("   " + "  ".join("%s %.1e" % (n.split(".", 1)[1][:14] if "user" in n else n.replace("news_encoder.", "n.")[:14], rel) for n, mag, rel in row))
