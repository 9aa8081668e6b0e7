import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model, fwd_bwd
from oracle import nrms_oracle as orc
shape = synth.Shape(n_words=300, word_embed_size=120, num_attention_heads=6, query_vector_dim=64, batch_size=12,
                    history_len=20, n_candidates=4, n_words_title=17)
kw = dict(seed=102, ragged=True, min_title=1, empty_history_user=True, all_pad_title=True, mask_some_candidates=True)
params = synth.make_params(shape, seed=101, pad_row_zero=False)
batch = synth.make_batch(shape, **kw)
model = make_model(shape, params, precision="fp16").train()
scores, loss, grads = fwd_bwd(model, batch)
o_scores, o_loss, o_grads, _ = orc.loss_and_grads(params, batch, shape.num_attention_heads)
for n, g in grads.items():
    bad = ~np.isfinite(g)
    print("%-62s nan/inf %d / %d   max|ref| %.2e  err(finite) %.2e" % (n, bad.sum(), g.size, np.abs(o_grads[n]).max(),
          np.abs(np.where(bad, 0, g) - np.where(bad, 0, o_grads[n])).max()))
    if bad.any() and g.ndim == 2:
        rows = np.unique(np.argwhere(bad)[:, 0])
        print("   bad rows:", rows[:30], " cols of first bad row:", np.argwhere(bad[rows[0]])[:10].ravel())
ids = np.concatenate([batch["browsed_titles"].reshape(-1), batch["candidate_titles"].reshape(-1)])
print("token count of id 0:", (ids == 0).sum(), " total", ids.size)
