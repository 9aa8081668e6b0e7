"""Localises errors of the fp16 fused forward: compares x16 / ctx16 / T16 / w / out of one encoder pass with a
torch fp32 restatement of the same stage.  GPU box only (diagnostic, not a test)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model

def p16_unpermute(a, h, dk):
    """[rows, 32h] in P16 order -> [rows, h*dk] model columns"""
    rows, DP = a.shape
    pos = np.arange(DP)
    b16, t = pos >> 4, pos & 15
    hh, j = t >> 3, t & 7
    fpad = 16 * b16 + 8 * (j >> 2) + 4 * hh + (j & 3)
    nat = np.zeros_like(a)
    nat[:, fpad] = a
    return nat.reshape(rows, DP // 32, 32)[:, :h, :dk].reshape(rows, h * dk)

def run(shape, seed=1, pad_zero=True, S_user=False):
    params = synth.make_params(shape, seed=101, pad_row_zero=pad_zero)
    batch = synth.make_batch(shape, seed=102, ragged=True, min_title=1)
    model = make_model(shape, params, precision="fp16").eval()
    eng = model.engine
    eng.fp16_backward = True                      # forward only: keeps t16 / w
    d, h, q, L = shape.word_embed_size, shape.num_attention_heads, shape.query_vector_dim, shape.n_words_title
    dk = d // h
    ids = torch.from_numpy(batch["candidate_titles"].reshape(-1, L)).cuda()
    N = ids.shape[0]
    out = eng.encode_titles(model._flat, ids, save=True, tag="dbg").cpu().numpy()
    torch.cuda.synchronize()
    KP, DP, QP = 320, 320, 224
    M = N * L
    # fragment order [title][k-step][token 0..31][16] -> token-major rows of the real tokens
    ctx16 = eng._bufs["dbg.ctx16"][:N * 32 * DP].view(N, DP // 16, 32, 16).permute(0, 2, 1, 3).reshape(N, 32, DP)[:, :L]
    ctx16 = ctx16.reshape(M, DP).float().cpu().numpy()
    t16 = eng._bufs["dbg.t16"][:N * 32 * QP].view(N, QP // 16, 32, 16).permute(0, 2, 1, 3).reshape(N, 32, QP)[:, :L]
    t16 = t16.reshape(M, QP).float().cpu().numpy()
    w = eng._bufs["dbg.w"][:M].cpu().numpy()
    # torch restatement
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    X = F.embedding(ids.cpu(), p["news_encoder.word_embedding.0.weight"])
    pre = "news_encoder.multihead_self_attention."
    def proj(n): return F.linear(X, p[pre + n + ".weight"], p[pre + n + ".bias"]).view(N, L, h, dk).transpose(1, 2)
    Q, K, V = proj("W_Q"), proj("W_K"), proj("W_V")
    A = F.softmax(Q @ K.transpose(-1, -2) / math.sqrt(dk), -1)
    ctx = (A @ V).transpose(1, 2).reshape(N, L, d)
    T = torch.tanh(F.linear(ctx, p["news_encoder.additive_attention.linear.weight"], p["news_encoder.additive_attention.linear.bias"]))
    sc = T @ p["news_encoder.additive_attention.attention_query_vector"]
    ww = F.softmax(sc, 1)
    o = torch.bmm(ww.unsqueeze(1), ctx).squeeze(1)
    got_ctx = p16_unpermute(ctx16, h, dk)
    e_ctx = np.abs(got_ctx - ctx.reshape(M, d).numpy())
    e_t = np.abs(t16[:, :q] - T.reshape(M, q).numpy())
    e_w = np.abs(w - ww.reshape(M).numpy())
    e_o = np.abs(out - o.numpy())
    print("shape d=%d h=%d q=%d L=%d N=%d: ctx %.3e  T %.3e  w %.3e  out %.3e" % (d, h, q, L, N, e_ctx.max(), e_t.max(), e_w.max(), e_o.max()))
    if e_ctx.max() > 1e-2:
        bad = np.argwhere(e_ctx > 1e-2)
        print("  ctx bad rows (token idx) sample:", np.unique(bad[:, 0])[:20], " bad cols sample:", np.unique(bad[:, 1])[:40])
        print("  tokens-per-title L=%d; bad title idx:" % L, np.unique(bad[:, 0] // L)[:20], "bad pos in title:", np.unique(bad[:, 0] % L)[:20])
        r = bad[0, 0]
        print("  row", r, "got", got_ctx[r, :12], "\n        want", ctx.reshape(M, d).numpy()[r, :12])
    if e_t.max() > 1e-2:
        bad = np.argwhere(e_t > 1e-2)
        print("  T bad rows:", np.unique(bad[:, 0])[:20], "cols:", np.unique(bad[:, 1])[:40])

run(synth.Shape(n_words=64, word_embed_size=8, num_attention_heads=2, query_vector_dim=4, batch_size=1, history_len=1, n_candidates=1, n_words_title=1))
run(synth.Shape(n_words=64, word_embed_size=8, num_attention_heads=2, query_vector_dim=4, batch_size=1, history_len=1, n_candidates=1, n_words_title=3))
run(synth.Shape(n_words=64, word_embed_size=32, num_attention_heads=1, query_vector_dim=32, batch_size=1, history_len=1, n_candidates=1, n_words_title=5))
run(synth.Shape(n_words=64, word_embed_size=32, num_attention_heads=1, query_vector_dim=32, batch_size=3, history_len=1, n_candidates=4, n_words_title=32))
run(synth.Shape(n_words=300, word_embed_size=60, num_attention_heads=6, query_vector_dim=32, batch_size=5, history_len=9, n_candidates=4, n_words_title=11))
run(synth.Shape(n_words=300, word_embed_size=300, num_attention_heads=10, query_vector_dim=200, batch_size=4, history_len=9, n_candidates=5, n_words_title=30))
run(synth.Shape(n_words=300, word_embed_size=300, num_attention_heads=10, query_vector_dim=200, batch_size=4, history_len=9, n_candidates=5, n_words_title=30), pad_zero=False)
