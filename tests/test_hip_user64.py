"""The user encoder as one kernel per direction (csrc/user64.hip, NRMS_FLAG_FUSED_SEQ64: UserEncoder.forward,
model/nrms_v0.py:188-199, and its autograd, in split-bf16) against the ORACLE -- user vectors, d(news vectors) and all nine
parameter gradients through autograd on the oracle's user_encoder -- and against the unfused bf16x3 chain it replaces, on the
benchmarked geometry and on the edges of what the kernels accept (33 and 64 rows, odd user counts, heads of 10 / 30 / 32 columns,
fewer than ten heads, q below 224)."""
import ctypes as C

import numpy as np
import pytest
import torch

from pytorch_news_recommender_amd import _lib, synth
from tests.test_hip_parity import make_model

pytestmark = pytest.mark.gpu

CASES = {
    # B, H, d, heads, q
    "bench": (64, 50, 300, 10, 200),
    "rows33_odd_users": (7, 33, 300, 10, 200),
    "rows64": (5, 64, 300, 10, 200),
    "dk10_h6": (9, 40, 60, 6, 32),
    "dk32_h4": (6, 50, 128, 4, 224),
    "dk2_h6": (3, 34, 12, 6, 4),
    "one_user": (1, 50, 300, 10, 200),
}


def _setup(case):
    B, H, d, h, q = CASES[case]
    shape = synth.Shape(n_words=50, word_embed_size=d, num_attention_heads=h, query_vector_dim=q, batch_size=B,
                        history_len=H, n_candidates=2, n_words_title=4)
    params = synth.make_params(shape, seed=17)
    g = torch.Generator().manual_seed(18)
    x = torch.randn(B, H, d, generator=g) * 0.3
    dout = torch.randn(B, d, generator=g) * 1e-2
    return shape, params, x, dout


def _hip(model, x, dout, fused):
    eng, flat = model.engine, model._flat
    eng.fused_user_encoder = fused
    desc = eng._desc("user_encoder", x.shape[0], x.shape[1], training=True)
    assert bool(desc.flags & _lib.NRMS_FLAG_FUSED_SEQ64) is fused
    xd, dd = x.cuda(), dout.cuda()
    out = eng.encode_users(flat, xd, save=True, tag="t64").clone()
    inf = eng.encode_users(flat, xd, tag="t64i").clone()
    gf = torch.zeros_like(flat)
    dx = eng.encode_users_backward(flat, gf, xd, dd, tag="t64").clone()
    torch.cuda.synchronize()
    grads = {n: model._layout.view(gf, n).cpu().numpy().copy() for n in model._layout.names if n.startswith("user_encoder.")}
    return out.cpu().numpy(), inf.cpu().numpy(), dx.cpu().numpy(), grads


@pytest.mark.parametrize("case", sorted(CASES))
def test_fused_user_encoder_against_the_oracle_and_the_chain(case):
    from oracle import nrms_oracle as orc
    shape, params, x, dout = _setup(case)
    # the oracle: user vectors, and through autograd d(x) and the parameter gradients of <out, dout>
    pt = orc.to_torch({k: v for k, v in params.items() if k.startswith("user_encoder.")}, requires_grad=True)
    xo = x.clone().requires_grad_(True)
    o_out = orc.user_encoder(pt, xo, shape.num_attention_heads)
    (o_out * dout).sum().backward()
    model = make_model(shape, params, precision="bf16x3")
    out, inf, dx, grads = _hip(model, x, dout, True)
    c_out, _, c_dx, c_grads = _hip(model, x, dout, False)
    scale = float(o_out.detach().abs().max())
    err, cerr = float(np.abs(out - o_out.detach().numpy()).max()), float(np.abs(c_out - o_out.detach().numpy()).max())
    print("user64 %-18s out: fused %.2e  chain %.2e (scale %.2f)" % (case, err, cerr, scale))
    assert err < 2e-5 * max(1.0, scale)
    assert float(np.abs(inf - out).max()) <= 2e-7 * max(1.0, scale)       # the inference instantiation (no saved activations)
    dxe = float(np.abs(dx - xo.grad.numpy()).max())
    dxs = float(xo.grad.abs().max())
    assert dxe <= 1e-3 * dxs + 2e-9, (dxe, dxs)
    gscale = max(float(np.abs(pt[n].grad.numpy()).max()) for n in grads)
    for n, g in grads.items():
        ref = pt[n].grad.numpy()
        bound = 1e-3 * np.abs(ref) + 2e-6 * gscale + 2e-5 * float(np.abs(ref).max()) + 1e-9
        bad = np.abs(g - ref) - bound
        cbad = float(np.abs(c_grads[n] - ref).max())
        print("      %-62s fused err %.2e  chain err %.2e  scale %.2e" % (n, float(np.abs(g - ref).max()), cbad, float(np.abs(ref).max())))
        assert float(bad.max()) <= 0.0, (case, n, float(np.abs(g - ref).max()), float(np.abs(ref).max()))


def test_fused_user_encoder_is_bit_reproducible_and_users_are_independent():
    shape, params, x, dout = _setup("bench")
    model = make_model(shape, params, precision="bf16x3")
    a = _hip(model, x, dout, True)
    b = _hip(model, x, dout, True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    for n in a[3]:
        assert np.array_equal(a[3][n], b[3][n]), n
    # a user's vector and input gradient do not depend on its position in the batch or on its workgroup partner
    perm = torch.from_numpy(np.random.default_rng(2).permutation(x.shape[0]))
    c = _hip(model, x[perm], dout[perm], True)
    assert np.array_equal(c[0], a[0][perm.numpy()]) and np.array_equal(c[2], a[2][perm.numpy()])


def test_fused_flag_is_refused_outside_its_shapes():
    lib = _lib.load()
    base = dict(n_seq=4, seq_len=50, d_model=300, n_heads=10, q_dim=200, vocab=0, p_drop_embed=0.0, p_drop_ctx=0.0,
                precision=_lib.NRMS_PRECISION_BF16X3, use_output_proj=0, mask_mode=0, flags=_lib.NRMS_FLAG_FUSED_SEQ64, seed=0,
                loss_scale=0.0, p_drop_attn=0.0)
    ok = _lib.EncoderDesc(**base)
    assert lib.nrms_encoder_fused_qkv_bytes(C.byref(ok)) == 4 * 10 * 2 * 12 * 1024 + 4 * 2 * 20 * 2048
    assert lib.nrms_encoder_fwd_scratch_bytes(C.byref(ok)) > 37 * 40960
    for kw in (dict(seq_len=32), dict(seq_len=20), dict(d_model=304, n_heads=8), dict(n_heads=5), dict(q_dim=256), dict(use_output_proj=1),
               dict(mask_mode=2), dict(precision=_lib.NRMS_PRECISION_FP32), dict(vocab=100), dict(p_drop_ctx=0.1)):
        args = dict(base)
        args.update(kw)
        bad = _lib.EncoderDesc(**args)
        assert lib.nrms_encoder_fused_qkv_bytes(C.byref(bad)) == 0, kw
        assert lib.nrms_encoder_fwd_scratch_bytes(C.byref(bad)) == 0, kw
        assert b"FUSED_SEQ64" in lib.nrms_last_error(), (kw, lib.nrms_last_error())
