"""Does a user's score depend on its batch position in the fp16 mode?  (bit-exact check + where the differences are)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_news_recommender_amd import synth
from tests.test_hip_parity import make_model
shape = synth.BENCH
params = synth.make_params(shape, seed=0)
batch = synth.make_batch(shape, seed=1, mask_some_candidates=True)
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
model = make_model(shape, params, precision="fp16")
eng = model.engine
def sc(t, training=True):
    return eng.forward(model._flat, t["browsed_titles"], t["candidate_titles"], t["candidate_mask"], training=training).clone()
for training in (True, False):
    s1 = sc(tb, training); s2 = sc(tb, training)
    print("training=%s rerun equal: %s" % (training, torch.equal(s1, s2)))
    perm = torch.from_numpy(np.random.default_rng(4).permutation(shape.batch_size)).cuda()
    sp = sc({k: v[perm] for k, v in tb.items()}, training)
    d = (sp - s1[perm]).abs()
    bad = (d > 0).any(dim=1)
    print("   permuted: equal %s; users differing %d / %d; max |diff| %.3e" % (torch.equal(sp, s1[perm]), int(bad.sum()), len(bad), float(d.max())))
    # news vectors alone
    B, H, L = tb["browsed_titles"].shape
    ids = tb["browsed_titles"].reshape(B * H, L)
    v1 = eng.encode_titles(model._flat, ids, chunk_titles=1 << 20).clone()
    p2 = torch.from_numpy(np.random.default_rng(5).permutation(B * H)).cuda()
    v2 = eng.encode_titles(model._flat, ids[p2].contiguous(), chunk_titles=1 << 20)
    dv = (v2 - v1[p2]).abs().max(dim=1).values
    nl = (ids[p2] != 0).sum(1)
    print("   news vectors under a title permutation: differing titles %d / %d, max %.3e" % (int((dv > 0).sum()), len(dv), float(dv.max())))
    for cls, name in ((nl == 0, "all-padding"), ((nl > 0) & (nl <= 15), "short"), (nl > 15, "long")):
        print("      %-12s differing %d / %d" % (name, int(((dv > 0) & cls).sum()), int(cls.sum())))

# ---- where do the differing titles sit?
B, H, L = tb["browsed_titles"].shape
ids = tb["browsed_titles"].reshape(B * H, L)
v1 = eng.encode_titles(model._flat, ids, chunk_titles=1 << 20).clone()
p2 = torch.from_numpy(np.random.default_rng(5).permutation(B * H)).cuda()
ids2 = ids[p2].contiguous()
v2 = eng.encode_titles(model._flat, ids2, chunk_titles=1 << 20)
dv = (v2 - v1[p2]).abs().max(dim=1).values.cpu().numpy()
def layout(idm):
    idm = idm.cpu().numpy()
    n = (idm != 0).sum(1)
    prefix = np.array([(row[:k] != 0).all() for row, k in zip(idm, n)])
    short = np.nonzero((n > 0) & (n <= 15) & prefix)[0]
    rank = {int(t): r for r, t in enumerate(short)}
    return n, short, rank
nA, shortA, rankA = layout(ids)
nB, shortB, rankB = layout(ids2)
p2c = p2.cpu().numpy()
for t in np.nonzero(dv > 0)[0]:
    orig = int(p2c[t])
    rb, ra = rankB[int(t)], rankA[orig]
    def info(rank, short, n):
        partner = rank ^ 1
        pn = int(n[short[partner]]) if partner < len(short) else -1
        return "rank %5d half %d wave %d wg %4d partner_n %2d" % (rank, rank & 1, (rank // 2) % 4, rank // 8, pn)
    print("title n=%2d diff %.2e | permuted run: %s | original run: %s | n_short %d" % (int(nB[t]), dv[t], info(rb, shortB, nB), info(ra, shortA, nA), len(shortB)))

# ---- which activations differ?  (ctx16 fragment blocks and pooling weights of the differing titles)
def acts_of(idm):
    out = eng.encode_titles(model._flat, idm, chunk_titles=1 << 20, save=True, tag="dbg")
    torch.cuda.synchronize()
    n_t = idm.shape[0]
    ctx = eng._bufs["dbg.ctx16"][: n_t * 32 * 320].view(n_t, 20, 32, 16).float().cpu().numpy().copy()
    w = eng._bufs["dbg.w"][: n_t * L].view(n_t, L).cpu().numpy().copy()
    return out.cpu().numpy(), ctx, w
o1, c1, w1 = acts_of(ids)
o2, c2, w2 = acts_of(ids2)
for t in np.nonzero(dv > 0)[0]:
    orig = int(p2c[t])
    dc = np.abs(c2[t] - c1[orig])
    ks, tok, e = np.nonzero(dc)
    print("title (perm idx %d): ctx16 elements differing %d of %d; k-steps %s tokens %s max %.3e; w max diff %.3e" % (
        t, len(ks), dc.size, sorted(set(ks.tolist())), sorted(set(tok.tolist())), dc.max(), np.abs(w2[t] - w1[orig]).max()))
