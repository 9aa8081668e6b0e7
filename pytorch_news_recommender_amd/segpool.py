"""Host side of the gather + additive-attention aggregate (``nrms_segment_pool_fwd`` / ``_bwd``, csrc/segpool.hip): SURVEY
section 8 row f-4 -- the aggregation step of a HieRec-style hierarchical interest model and of a user-news graph encoder.
The reference has no implementation of either (``model/tanr.py`` is empty): PARITY UNPINNED (checked against
``oracle/segpool_oracle.py``, a restatement of the reference's own additive attention over an index list).

``SegmentPool`` owns the buffers of one call site (via torch: allocator + streams only) and issues the two C-ABI calls;
``segment_pool`` is the autograd form.  No CPU fallback: without the library or a GPU this raises."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SegmentPool:
    def __init__(self, d, q, precision="bf16x3", rows_unique=False):
        self.lib = _lib.load()
        if precision not in ("fp32", "bf16x3", "bf16"):
            raise ValueError("precision must be fp32, bf16x3 or bf16")
        self.d, self.q, self.precision, self.rows_unique = int(d), int(q), precision, bool(rows_unique)
        self._ws = None
        self._saved = None

    def _desc(self, n_rows, n_seg, nnz):
        return _lib.SegPoolDesc(n_rows=n_rows, n_seg=n_seg, nnz=nnz, d=self.d, q=self.q, precision=_lib.PRECISIONS[self.precision],
                                flags=_lib.NRMS_SEGPOOL_ROWS_UNIQUE if self.rows_unique else 0)

    def _workspace(self, desc, device):
        nbytes = int(self.lib.nrms_segment_pool_workspace_bytes(C.byref(desc)))
        if nbytes == 0 and self.lib.nrms_last_error():
            _lib.check(-1, "nrms_segment_pool_workspace_bytes")
        if self._ws is None or self._ws.numel() * 4 < nbytes or self._ws.device != device:
            self._ws = torch.empty((nbytes + 3) // 4 + 64, dtype=torch.float32, device=device)
        return self._ws

    def forward(self, x, w_add, b_add, q_vec, seg_ptr, idx):
        """x [R, d] fp32, seg_ptr int32 [n_seg + 1], idx int32 [nnz] (device) -> out [n_seg, d]."""
        if x.device.type != "cuda":
            raise _lib.NrmsError("segment_pool needs GPU tensors (there is no CPU path)")
        x = x.contiguous()
        R, n_seg, nnz = x.shape[0], seg_ptr.numel() - 1, idx.numel()
        desc = self._desc(R, n_seg, nnz)
        ws = self._workspace(desc, x.device)
        t = torch.empty(R, self.q, dtype=torch.float32, device=x.device)
        logit = torch.empty(max(R, 1), dtype=torch.float32, device=x.device)
        alpha = torch.empty(max(nnz, 1), dtype=torch.float32, device=x.device)
        out = torch.empty(n_seg, self.d, dtype=torch.float32, device=x.device)
        rc = self.lib.nrms_segment_pool_fwd(C.byref(desc), _lib.ptr(x), _lib.ptr(w_add.contiguous()), _lib.ptr(b_add.contiguous()),
                                            _lib.ptr(q_vec.contiguous()), _lib.ptr(seg_ptr), _lib.ptr(idx), _lib.ptr(t), _lib.ptr(logit),
                                            _lib.ptr(alpha), _lib.ptr(out), _lib.ptr(ws), C.c_size_t(ws.numel() * 4), _stream())
        _lib.check(rc, "nrms_segment_pool_fwd")
        self._saved = (x, seg_ptr, idx, t, alpha, desc)
        return out

    def backward(self, w_add, q_vec, dout, dw_add, db_add, dq_vec):
        """dout [n_seg, d] -> dx [R, d]; dw_add / db_add / dq_vec (fp32, contiguous) are accumulated."""
        x, seg_ptr, idx, t, alpha, desc = self._saved
        ws = self._workspace(desc, x.device)
        dx = torch.empty_like(x)
        rc = self.lib.nrms_segment_pool_bwd(C.byref(desc), _lib.ptr(x), _lib.ptr(w_add.contiguous()), _lib.ptr(q_vec.contiguous()),
                                            _lib.ptr(seg_ptr), _lib.ptr(idx), _lib.ptr(t), _lib.ptr(alpha), _lib.ptr(dout.contiguous()),
                                            _lib.ptr(dx), _lib.ptr(dw_add), _lib.ptr(db_add), _lib.ptr(dq_vec), _lib.ptr(ws),
                                            C.c_size_t(ws.numel() * 4), _stream())
        _lib.check(rc, "nrms_segment_pool_bwd")
        return dx


class _SegmentPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_add, b_add, q_vec, seg_ptr, idx, precision, rows_unique):
        op = SegmentPool(x.shape[1], w_add.shape[0], precision, rows_unique)
        out = op.forward(x.detach(), w_add.detach(), b_add.detach(), q_vec.detach(), seg_ptr, idx)
        ctx.op = op
        ctx.save_for_backward(w_add, q_vec)
        return out

    @staticmethod
    def backward(ctx, dout):
        w_add, q_vec = ctx.saved_tensors
        dw, db, dq = torch.zeros_like(w_add), torch.zeros(w_add.shape[0], device=w_add.device), torch.zeros_like(q_vec)
        dx = ctx.op.backward(w_add.detach(), q_vec.detach(), dout, dw, db, dq)
        return dx, dw, db, dq, None, None, None, None


def segment_pool(x, w_add, b_add, q_vec, seg_ptr, idx, precision="bf16x3", rows_unique=False):
    """Differentiable (x, w_add, b_add, q_vec): out[s] = sum_k softmax_k(q . tanh(W x_k + b)) x_k over the members of segment s."""
    return _SegmentPoolFn.apply(x, w_add, b_add, q_vec, seg_ptr, idx, precision, rows_unique)
