"""Gather + additive-attention aggregate (csrc/segpool.hip; SURVEY f-4) against its torch restatement (oracle/segpool_oracle.py).
PARITY UNPINNED: the reference holds no implementation of the HieRec / graph-encoder models this serves (model/tanr.py is
empty) -- the oracle restates the reference's additive attention (nrms_v0.py:100-126) over an index list."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(R, n_seg, d, q, seed, partition, empty_some=True, max_len=40):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((R, d)) * 0.5).astype(np.float32)
    w = (rng.uniform(-1, 1, (q, d)) * np.sqrt(6.0 / (q + d))).astype(np.float32)
    b = (rng.uniform(-0.05, 0.05, q)).astype(np.float32)
    qv = rng.uniform(-0.1, 0.1, q).astype(np.float32)
    if partition:
        owner = rng.integers(0, n_seg + (2 if empty_some else 0), R)          # some rows belong to nobody, some segments are empty
        idx, ptr = [], [0]
        for s in range(n_seg):
            m = np.nonzero(owner == s)[0]
            rng.shuffle(m)
            idx += m.tolist()
            ptr.append(len(idx))
    else:
        idx, ptr = [], [0]
        for s in range(n_seg):
            n = 0 if (empty_some and s % 7 == 3) else int(rng.integers(1, max_len))
            idx += rng.integers(0, R, n).tolist()                             # repeats across (and inside) segments
            ptr.append(len(idx))
    return x, w, b, qv, np.asarray(ptr, np.int32), np.asarray(idx, np.int32)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", ["partition_300", "graph_300", "partition_odd", "graph_wide", "one_row"])
def test_segment_pool_forward_and_gradients_against_the_oracle(case, precision):
    from oracle import segpool_oracle as orc
    from pytorch_news_recommender_amd.segpool import segment_pool
    cfg = {"partition_300": (400, 60, 300, 200, True), "graph_300": (300, 90, 300, 200, False), "partition_odd": (77, 13, 20, 8, True),
           "graph_wide": (64, 20, 800, 400, False), "one_row": (1, 1, 300, 200, True)}[case]
    R, n_seg, d, q, partition = cfg
    x, w, b, qv, ptr, idx = _case(R, n_seg, d, q, seed=5, partition=partition, empty_some=case != "one_row")
    if case == "one_row":
        ptr, idx = np.asarray([0, 1], np.int32), np.asarray([0], np.int32)
    to = lambda a: torch.from_numpy(a).clone().requires_grad_(True)
    ox, ow, ob, oq = to(x), to(w), to(b), to(qv)
    o_out = orc.segment_pool(ox, ow, ob, oq, ptr.tolist(), idx.tolist())
    g = torch.from_numpy(np.random.default_rng(9).standard_normal(o_out.shape).astype(np.float32) * 1e-2)
    (o_out * g).sum().backward()
    dev = torch.device("cuda")
    hx, hw, hb, hq = (torch.from_numpy(a).to(dev).requires_grad_(True) for a in (x, w, b, qv))
    out = segment_pool(hx, hw, hb, hq, torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev), precision=precision, rows_unique=partition)
    (out * g.to(dev)).sum().backward()
    tol = 1e-5 if precision == "fp32" else 2e-5
    err = float((out.detach().cpu() - o_out.detach()).abs().max())
    print("segpool %-14s %-6s out err %.2e (scale %.2f)" % (case, precision, err, float(o_out.abs().max())))
    assert err < tol * max(1.0, float(o_out.abs().max()))
    for name, h, o in (("dx", hx, ox), ("dW", hw, ow), ("db", hb, ob), ("dq", hq, oq)):
        ref = o.grad
        bound = 1e-3 * ref.abs() + 2e-5 * float(ref.abs().max()) + 1e-9
        diff = (h.grad.cpu() - ref).abs()
        print("      %-3s err %.2e scale %.2e" % (name, float(diff.max()), float(ref.abs().max())))
        assert bool((diff <= bound).all()), (case, precision, name, float(diff.max()), float(ref.abs().max()))


@pytest.mark.parametrize("partition", [True, False], ids=["partition", "shared_rows"])
def test_both_modes_are_bit_reproducible(partition):
    """Segments that partition the rows (plain stores) and rows listed by many segments (entries sorted by row, added in list order):
    the same bits on every run, whatever order the hardware schedules the waves in."""
    from pytorch_news_recommender_amd.segpool import SegmentPool
    x, w, b, qv, ptr, idx = _case(500, 70, 300, 200, seed=11, partition=partition)
    dev = torch.device("cuda")
    tx, tw, tb, tq, tp, ti = (torch.from_numpy(a).to(dev) for a in (x, w, b, qv, ptr, idx))
    dout = torch.randn(70, 300, generator=torch.Generator().manual_seed(3)).to(dev)
    res = []
    for _ in range(3):
        op = SegmentPool(300, 200, "bf16x3", rows_unique=partition)
        out = op.forward(tx, tw, tb, tq, tp, ti).clone()
        dw, db, dq = torch.zeros_like(tw), torch.zeros_like(tb), torch.zeros_like(tq)
        dx = op.backward(tw, tq, dout, dw, db, dq)
        torch.cuda.synchronize()
        res.append((out, dx.clone(), dw, db, dq))
    for other in res[1:]:
        for a, c in zip(res[0], other):
            assert torch.equal(a, c)


def test_descriptor_validation():
    import ctypes as C
    from pytorch_news_recommender_amd import _lib
    lib = _lib.load()
    for kw in (dict(d=301), dict(q=6), dict(d=2048), dict(precision=_lib.NRMS_PRECISION_FP16), dict(flags=8), dict(n_rows=-1)):
        args = dict(n_rows=10, n_seg=2, nnz=4, d=300, q=200, precision=_lib.NRMS_PRECISION_BF16X3, flags=0)
        args.update(kw)
        assert lib.nrms_segment_pool_workspace_bytes(C.byref(_lib.SegPoolDesc(**args))) == 0, kw
