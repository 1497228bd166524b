"""Poincare-ball math on HIP: same function names / argument meaning as hyptorch/pmath.py (forward values only).

Every function takes CUDA(HIP) fp32 tensors and raises on CPU tensors (no fallback).  Leading dims are flattened to
rows; the last dim is the feature dim.  ``auto_select_c`` is host arithmetic (scipy gamma, pmath.py:496-505).
"""
import numpy as np
import torch

from . import capi

_OPS = dict(project=0, lambda_x=1, mobius_add=2, dist=3, dist0=4, expmap=5, expmap0=6, logmap=7, logmap0=8, p2k=9, k2p=10,
            lorenz=11, oblique_proj=12)


def _prep(x):
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise capi.SttodeError('sttode_amd.pmath runs only on HIP tensors (no CPU fallback)')
    return x.to(torch.float32).contiguous()


def _row(op, x, y=None, c=1.0, scalar=False, keepdim=False):
    x = _prep(x)
    if y is not None:
        y = _prep(y)
        x, y = torch.broadcast_tensors(x, y)
        x, y = x.contiguous(), y.contiguous()
    d = x.shape[-1]
    rows = x.numel() // d
    out = torch.empty(rows if scalar else (rows, d), dtype=torch.float32, device=x.device)
    capi.call('sttode_pmath_rowop', _OPS[op], x, y, out, None, rows, d, float(c), capi.stream_ptr())
    if scalar:
        return out.view(*x.shape[:-1], 1) if keepdim else out.view(*x.shape[:-1])
    return out.view(x.shape)


def _scalar(which, x):
    x = _prep(x)
    out = torch.empty_like(x)
    capi.call('sttode_pmath_scalar', which, x, out, x.numel(), capi.stream_ptr())
    return out


def tanh(x, clamp=15):
    assert clamp == 15
    return _scalar(0, x)


class Artanh(torch.autograd.Function):
    """hyptorch/pmath.py:16-27 (forward clamps to +-(1 - 1e-5); backward grad / (1 - x_clamped^2))."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _scalar(1, x)

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        return _mul(grad_output, _scalar(3, x))


class Arsinh(torch.autograd.Function):
    """hyptorch/pmath.py:51-60."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _scalar(2, x)

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        return _mul(grad_output, _scalar(4, x))


class RiemannianGradient(torch.autograd.Function):
    """hyptorch/pmath.py:30-45: identity forward, gradient rescaled by (1 - c |x|^2)^2 / 4 (class attribute ``c``)."""

    c = 1

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        xs, g = _prep(x), _prep(grad_output)
        d = xs.shape[-1]
        out = torch.empty_like(g)
        capi.call('sttode_pmath_riemannian_grad', xs, g, out, xs.numel() // d, d, float(RiemannianGradient.c), capi.stream_ptr())
        return out.view_as(grad_output)


def _mul(a, b):
    a, b = _prep(a), _prep(b)
    out = torch.empty_like(a)
    capi.call('sttode_train_ewise', 0, out, a, b, None, None, out.numel(), 0, 0.0, capi.stream_ptr())
    return out


def artanh(x):
    return Artanh.apply(x)


def arsinh(x):
    return Arsinh.apply(x)


def project(x, *, c=1.0):
    return _row('project', x, c=c)


def lambda_x(x, *, c=1.0, keepdim=False):
    return _row('lambda_x', x, c=c, scalar=True, keepdim=keepdim)


def mobius_add(x, y, *, c=1.0):
    return _row('mobius_add', x, y, c=c)


def dist(x, y, *, c=1.0, keepdim=False):
    return _row('dist', x, y, c=c, scalar=True, keepdim=keepdim)


def dist0(x, *, c=1.0, keepdim=False):
    return _row('dist0', x, c=c, scalar=True, keepdim=keepdim)


def expmap(x, u, *, c=1.0):
    return _row('expmap', x, u, c=c)


def expmap0(u, *, c=1.0):
    return _row('expmap0', u, c=c)


def logmap(x, y, *, c=1.0):
    return _row('logmap', x, y, c=c)


def logmap0(y, *, c=1.0):
    return _row('logmap0', y, c=c)


def p2k(x, c):
    return _row('p2k', x, c=c)


def k2p(x, c):
    return _row('k2p', x, c=c)


def lorenz_factor(x, *, c=1.0, dim=-1, keepdim=False):
    assert dim in (-1, x.dim() - 1)
    return _row('lorenz', x, c=c, scalar=True, keepdim=keepdim)


def mobius_matvec(m, x, *, c=1.0):
    m, x = _prep(m), _prep(x)
    O, d = m.shape
    rows = x.numel() // d
    mx = torch.empty(rows, O, dtype=torch.float32, device=x.device)
    xn = torch.empty(rows, dtype=torch.float32, device=x.device)
    out = torch.empty(rows, O, dtype=torch.float32, device=x.device)
    capi.call('sttode_pmath_matvec', m, x, mx, xn, out, rows, d, O, float(c), capi.stream_ptr())
    return out.view(*x.shape[:-1], O)


def _pair(which, x, y, A, c, shape):
    x, y = _prep(x), _prep(y)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    capi.call('sttode_pmath_pair', which, x, y, _prep(A) if A is not None else None, out, x.shape[0], y.shape[0], x.shape[1], float(c),
              capi.stream_ptr())
    return out


def dist_matrix(x, y, c=1.0):
    return _pair(0, x, y, None, c, (x.shape[0], y.shape[0]))


def _mobius_addition_batch(x, y, c):
    return _pair(1, x, y, None, float(c), (x.shape[0], y.shape[0], x.shape[1]))


def _hyperbolic_softmax(X, A, P, c):
    return _pair(2, P, X, A, float(c), (X.shape[0], P.shape[0]))


def poincare_mean(x, dim=0, c=1.0):
    assert dim == 0 and x.dim() == 2
    x = _prep(x)
    rows, d = x.shape
    yl = torch.empty_like(x)
    lam = torch.empty(rows, dtype=torch.float32, device=x.device)
    out = torch.empty(d, dtype=torch.float32, device=x.device)
    capi.call('sttode_pmath_mean', x, yl, lam, out, rows, d, float(c), capi.stream_ptr())
    return out


def auto_select_c(d):
    """Ball radius such that the d-dimensional ball has volume pi (host arithmetic, pmath.py:496-505)."""
    from scipy.special import gamma
    dim2 = d / 2.0
    R = gamma(dim2 + 1) / (np.pi ** (dim2 - 1))
    R = R ** (1 / float(d))
    return 1 / (R ** 2)


# Oblique manifold (core/manifolds/oblique.py)
def oblique_proj(p):
    return _row('oblique_proj', p)


def oblique_dist(p1, p2):
    """Oblique.dist(p1, p2): [..., n1, d], [..., n2, d] -> [..., n2, n1]."""
    p1, p2 = _prep(p1), _prep(p2)
    n1, n2, d = p1.shape[-2], p2.shape[-2], p1.shape[-1]
    nb = p1.numel() // (n1 * d)
    out = torch.empty(*p1.shape[:-2], n2, n1, dtype=torch.float32, device=p1.device)
    capi.call('sttode_oblique_dist', p1, p2, out, nb, n1, n2, d, capi.stream_ptr())
    return out
