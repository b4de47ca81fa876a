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


def assert_close_via_f64(got, ref32, ref64, name='', rel=1e-4):
    """The fp64 triangle: where the product adds its terms in another order than the fp32 reference,
    both are compared with the same computation in float64 and the product may be as far from it as the fp32 reference
    itself is -- or 1e-4 of the tensor's scale, whichever is larger:
        |got - f64| <= max(max|ref32 - f64|, rel * min(max|f64|, 1)) + rel * |f64|   element-wise."""
    got, ref32, ref64 = _np(got).astype(np.float64), _np(ref32).astype(np.float64), _np(ref64).astype(np.float64)
    scale = float(np.abs(ref64).max()) if ref64.size else 0.0
    assert scale > 0.0, f'{name}: the float64 reference is identically zero -- the comparison would check nothing'
    ref_err = float(np.abs(ref32 - ref64).max())
    allowed = max(ref_err, rel * min(scale, 1.0))
    err = np.abs(got - ref64)
    worst = float((err - rel * np.abs(ref64)).max())
    assert worst <= allowed, (f'{name}: |product - f64| peaks {float(err.max()):.3g} (scale {scale:.3g}); allowed '
                              f'{allowed:.3g} = max(fp32 reference\'s own error {ref_err:.3g}, {rel:g} x scale)')
    return float(err.max()), ref_err, scale

