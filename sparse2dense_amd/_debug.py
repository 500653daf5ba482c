"""Debugging stand-ins used by tools/side_stress.py (HISTORY.md section 7 (f), rule 36) - nothing here runs unless its environment switch is set.

S2D_DEBUG_CT_WGRAD=<mode> replaces the weight-gradient launch of the PCR head's narrow up-sampler (the 16 -> 3 ConvTranspose3d with the
input-norm fold, dense3d._ConvT3dFn.backward) by:
  zeros        one tiny kernel, no read of chain data
  read         torch reductions over the same operands, result unused
  long         ~0.3 ms of chip-filling elementwise kernels over the input
  lds          softmax / row sums over the input
  ldsfill[:v]  a kernel that only fills the LDS of every CU with v (default NaN) for a few ms (csrc/common.hip s2d_debug_lds_fill)
  privws       the real kernels with a workspace of their own
  clone        the real kernels on private copies of every operand, made on the stream the closure runs on
  clone_main   ... on copies made on the chain's stream before the closure
None of the stand-ins perturbs the chain; the real kernels do in every variant - which is how the search left the operands and went to the
victim kernel's instruction stream."""
import torch

from . import _lib
from .dense2d import _ptr, _stream


def ct_wgrad_stand_in(mode, dw, x, dout, in_norm, dims, clones):
    lib = _lib.load()
    n, cin, cout, d, h, w = dims
    if mode == "zeros":
        return dw.zero_()
    if mode.startswith("ldsfill"):
        dw.zero_()
        val = float(mode.split(":")[1]) if ":" in mode else float("nan")
        _lib.check(lib.s2d_debug_lds_fill(val, 1024, 2000, None, _stream()), "s2d_debug_lds_fill")
        return dw
    if mode == "lds":
        dw.zero_()
        t = x.view(-1, x.shape[-1])
        acc = None
        for _ in range(8):
            t2 = torch.softmax(t, -1)
            acc = t2.sum(-1) if acc is None else acc + t2.sum(-1)
        dw.view(-1)[0] = acc.sum()
        return dw
    if mode == "long":
        dw.zero_()
        t = x
        for _ in range(12):
            t = t * 1.0001
        dw.view(-1)[0] = t.sum()
        return dw
    if mode == "read":
        dw.zero_()
        dw.view(-1)[0] = x.sum() + dout.float().sum() + in_norm.sum()
        return dw
    if mode in ("clone", "clone_main", "privws"):
        if mode == "clone":
            x, dout, in_norm = x.clone(), dout.clone(), in_norm.clone()
        elif mode == "clone_main":
            x, dout, in_norm = clones
        ws = torch.empty(max(lib.s2d_convt3d_mfma_wgrad_workspace_bytes(n, cin, cout, d, h, w), 256), dtype=torch.uint8, device=x.device)
        _lib.check(lib.s2d_convt3d_mfma_wgrad_d16_norm(_ptr(x), _ptr(in_norm), _ptr(dout), n, cin, cout, d, h, w, _ptr(dw), _ptr(ws), ws.numel(), _stream()),
                   "s2d_convt3d_mfma_wgrad_d16_norm")
        return dw
    raise ValueError(f"S2D_DEBUG_CT_WGRAD={mode!r}: unknown stand-in")
