"""Scale-aware comparison of gradients (a blanket atol = 1e-4 is larger than many of the gradients
it was checking -- the 1/n_el-scaled loss gradients peak at 1e-5..1e-3 -- so a zero or wrong-sign tensor passed).

``assert_grad_close(got, ref)``: |got - ref| <= rel * min(max|ref|, 1) + rel * |ref| element-wise, and the reference
tensor must be non-trivial.  Two declared exceptions, each named by the caller:
* ``zero=True``: the reference gradient is identically zero by construction (a branch the loss does not read);
  the product must give exact zeros (or no gradient at all);
* ``cancels=floor``: the true gradient is zero and the reference holds only rounding residue below ``floor``
  (a conv bias in front of train-mode BatchNorm); the product's residue must stay below the same floor."""
import numpy as np
import torch


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def assert_grad_close(got, ref, name='', rel=1e-4, zero=False, cancels=None):
    ref = _np(ref)
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    if zero:
        assert scale == 0.0, f'{name}: declared zero but the reference peaks at {scale:g}'
        if got is not None:
            assert float(np.abs(_np(got)).max()) == 0.0, f'{name}: reference gradient is exactly zero, product is not'
        return
    assert got is not None, f'{name}: no gradient'
    got = _np(got)
    if cancels is not None:
        assert scale < cancels, f'{name}: declared rounding residue (< {cancels:g}) but the reference peaks at {scale:g}'
        assert float(np.abs(got).max()) < cancels, f'{name}: residue {float(np.abs(got).max()):g} above {cancels:g}'
        return
    assert scale > 0.0, f'{name}: the reference gradient is identically zero -- the comparison would check nothing'
    np.testing.assert_allclose(got, ref, atol=rel * min(scale, 1.0), rtol=rel, err_msg=f'{name} (scale {scale:.3g})')
