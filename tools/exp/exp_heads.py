"""Experiment: what would halving the per-head work of the fused news-encoder kernels buy?  Same batch, same fixed pitches
(KP = DP = 320, 7 additive tiles), 10 heads of 30 vs 5 heads of 32 (d = 160): the per-title cost that is NOT per head (additive
stage, pooling, stores) is identical, so t(10) - t(5) = five heads' worth of work.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

for d, h in ((300, 10), (160, 5)):
    shape = synth.Shape(n_words=45800, word_embed_size=d, num_attention_heads=h, query_vector_dim=200, batch_size=512,
                        history_len=50, n_candidates=5, n_words_title=30)
    params = synth.make_params(shape, seed=0)
    batch = synth.make_batch(shape, seed=1)
    tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    model = make_model(shape, params, dropout=0.2, precision="fp16").train()
    for _ in range(3):
        model.train_step(tb)
    torch.cuda.synchronize()
    eng = model.engine
    os.environ["NRMS_NO_SIDE_STREAMS"] = "1"
    eng.timing(True); eng.timing_reset()
    for _ in range(10):
        model.train_step(tb)
    torch.cuda.synchronize()
    print("d=%d h=%d:" % (d, h), "  ".join("%s %.3f" % (n, eng.timing_read(n)[0] / 10) for n in
          ("fused_fwd16", "fused_bwd16_pool", "fused_bwd16_attn", "dx_bwd", "dwqkv_bwd", "dwadd_bwd")), flush=True)
    eng.timing(False)
    os.environ.pop("NRMS_NO_SIDE_STREAMS")
