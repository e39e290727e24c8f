"""The arithmetic of csrc/loss.hpp:focal_term, restated in numpy float32 operation by operation, against the torch module
of odtk/loss.py (= reference odtk/loss.py:13-19) evaluated in float64 -- values, sums and gradients, gamma 2 and general.
(The kernel itself is compared on the GPU in tests/test_gpu_loss.py; this is the CPU-side proof that the symmetric form
`loss(x, t) = alpha_t * sigmoid(s)^gamma * softplus(s)`, s = x for t = 0 and -x for t = 1, IS the reference's expression.)"""
import numpy as np
import pytest
import torch

from odtk import loss as L

f = np.float32


def focal_term(x, positive, w_neg, w_pos, gamma, backward):
    x = x.astype(f)
    s = np.where(positive, -x, x)
    e = np.exp2((np.abs(s) * f(-1.4426950408889634)).astype(f)).astype(f)       # v_exp_f32
    d = (f(1) + e).astype(f)
    r = (f(1) / d).astype(f)                                                     # v_rcp_f32
    er = (e * r).astype(f)
    nonneg = s >= 0
    q = np.where(nonneg, r, er)
    ce = (np.log2(d).astype(f) * f(0.6931471805599453) + np.maximum(s, 0)).astype(f)   # v_log_f32, v_fma_f32
    w = np.where(positive, w_pos, w_neg).astype(f)
    with np.errstate(divide='ignore'):
        mod = (q * q).astype(f) if gamma == 2 else np.exp2((f(gamma) * np.maximum(np.log2(q).astype(f), f(-150))).astype(f)).astype(f)
    if not backward:
        return (w * mod * ce).astype(f)
    omq = np.where(nonneg, er, r)
    return (w * mod * ((f(2.0 if gamma == 2 else gamma) * omq) * ce + q)).astype(f)


@pytest.mark.parametrize('gamma', [2.0, 1.5, 0.5, 0.0])
def test_symmetric_focal_form_is_the_reference_expression(gamma):
    rng = np.random.default_rng(int(gamma * 10))
    n, alpha = 200000, 0.25
    x = np.concatenate([(rng.standard_normal(n) * 3 - 2), [0.0, -0.0, 30.0, -30.0, 88.0, -88.0, 1e-8, -120.0, 120.0]]).astype(f)   # |x| > 104: exp underflows to 0
    positive = np.concatenate([rng.random(n) < 0.05, [False, True, True, False, False, True, True, False, True]])
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = L.FocalLoss(alpha, gamma)(xt, torch.tensor(positive, dtype=torch.float64))
    ref.sum().backward()
    want, grad = ref.detach().numpy(), xt.grad.numpy()
    got = focal_term(x, positive, f(1 - alpha), f(alpha), gamma, False)
    got_grad = focal_term(x, positive, f(1 - alpha), f(-alpha), gamma, True)      # ds/dx = -1 for t = 1 folded into w_pos
    assert np.isfinite(got).all() and np.isfinite(got_grad).all()
    assert abs(got.astype(np.float64).sum() - want.sum()) <= 1e-7 * want.sum()                   # the kernel's bar: 1e-6
    assert np.abs(got - want).max() <= 2e-7 * want.max()                                          # elementwise, absolute
    # gamma < 1: autograd of `(1 - pt) ** gamma` at 1 - pt == 0 is inf * 0 = NaN in the reference itself (float64: |x| > ~37,
    # float32: ~17); the symmetric form has no such singularity and returns the limit, 0
    ok = np.isfinite(grad)
    assert ok.all() or gamma < 1
    assert np.abs(got_grad - grad)[ok].max() <= 2e-6 * np.abs(grad[ok]).max()                     # the kernel's bar: 1e-5
    assert np.all(got_grad[~ok] == 0)
