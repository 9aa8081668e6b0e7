"""fp16-mode gradients against the fp32 mode at the bench shape, tensor by tensor (max error relative to the tensor's scale
and relative rms error), at initialisation and after k fp32 train steps on a fixed batch (which moves the weights off the
symmetric initialisation), with the user encoder in bf16x3 (default) and in fp16.  Also: how much a 5e-4 relative
perturbation of the news vectors moves the fp32 mode's own gradients (the conditioning of each tensor).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

shape = synth.BENCH
params = synth.make_params(shape, seed=0)
batch = synth.make_batch(shape, seed=1)
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
batch2 = synth.make_batch(shape, seed=2)
tb2 = {k: torch.from_numpy(v).cuda() for k, v in batch2.items()}
model = make_model(shape, params)
model.config.learning_rate = 1e-3
lay = model._layout


def grads(prec, u16=False, on=tb2):
    model.config.precision = prec
    model.config.fp16_user_encoder = u16
    eng = model.engine
    s = eng.forward(model._flat, on["browsed_titles"], on["candidate_titles"], on["candidate_mask"], training=True)
    _, ds = eng.ce_loss(s, grad_scale=1.0 / shape.batch_size)
    g = torch.zeros_like(model._flat)
    eng.backward(model._flat, g, ds)
    return g


def short(n):
    return n.replace("multihead_self_attention.", "").replace("additive_attention.", "add.").replace("_encoder", "")


done = 0
for k in (0, 10, 40, 150):
    model.config.precision = "fp32"
    while done < k:
        model.train_step(tb)
        done += 1
    ref = grads("fp32")
    print("== after %d fp32 train steps on a fixed batch (gradients on a fresh batch)" % k)
    for label, prec, u16 in (("fp16 (user bf16x3)", "fp16", False), ("fp16 (user fp16)", "fp16", True), ("bf16x3", "bf16x3", False)):
        g = grads(prec, u16)
        print("  " + label)
        for n in lay.names:
            a, b = lay.view(g, n).double(), lay.view(ref, n).double()
            sc = float(b.abs().max())
            print("     %-44s scale %.2e  max err %.1e of scale  rel rms %.1e" % (short(n), sc, float((a - b).abs().max()) / (sc + 1e-300),
                  float(((a - b) ** 2).mean().sqrt() / ((b ** 2).mean().sqrt() + 1e-300))))
model.config.precision = "fp32"
model.config.fp16_user_encoder = False
