"""Is the fp16 backward exactly linear under power-of-two changes of d(scores)?  (device-side loss scale)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model
shape = synth.Shape(n_words=400, word_embed_size=60, num_attention_heads=6, query_vector_dim=32,
                    batch_size=512, history_len=9, n_candidates=4, n_words_title=12)
params = synth.make_params(shape, seed=7)
batch = synth.make_batch(shape, seed=8, ragged=True, min_title=1, mask_some_candidates=True)
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
for u16 in (False, True):
    model = make_model(shape, params, precision="fp16", fp16_user=u16)
    eng, flat, lay = model.engine, model._flat, model._layout
    def grads(mult):
        s = eng.forward(flat, tb["browsed_titles"], tb["candidate_titles"], tb["candidate_mask"], training=True)
        _, ds = eng.ce_loss(s, grad_scale=1.0 / 512)
        g = torch.zeros_like(flat)
        eng.backward(flat, g, ds * mult)
        return g
    g1, g1b, g2, g512 = grads(1.0), grads(1.0), grads(2.0), grads(512.0)
    for n in lay.names:
        a = lay.view(g1, n)
        print("user16=%s %-60s rerun %.2e  x2 %.2e  x512 %.2e  (scale %.2e)" % (u16, n, float((lay.view(g1b, n) - a).abs().max()),
              float((lay.view(g2, n) - 2 * a).abs().max()), float((lay.view(g512, n) - 512 * a).abs().max()), float(a.abs().max())))
