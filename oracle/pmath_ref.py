"""CPU oracle for the Poincare-ball primitives (TEST INFRASTRUCTURE ONLY).

Restates hyptorch/pmath.py (forward values only) in PyTorch-CPU fp32, keeping the
reference's exact epsilons / clamps.  Pinned by tests/golden/pmath.npz, generated
by calling the reference's own functions (tests/golden/make_golden.py).
None of these are called by the STTODE model path; they are a
stand-alone op library with op-level parity.
"""
import torch


def _c(c, x):
    return torch.as_tensor(c, dtype=x.dtype)


def tanh_clamped(x, clamp=15.0):  # pmath.py:11-12
    return x.clamp(-clamp, clamp).tanh()


def artanh(x):  # pmath.py:16-22
    x = x.clamp(-1 + 1e-5, 1 - 1e-5)
    return 0.5 * (torch.log(1 + x) - torch.log(1 - x))


def arsinh(x):  # pmath.py:51-55
    return (x + torch.sqrt(1 + x.pow(2))).clamp_min(1e-5).log()


def project(x, c=1.0):  # pmath.py:98-103
    c = _c(c, x)
    norm = x.norm(dim=-1, keepdim=True, p=2).clamp_min(1e-5)
    maxnorm = (1 - 1e-3) / (c ** 0.5)
    return torch.where(norm > maxnorm, x / norm * maxnorm, x)


def lambda_x(x, c=1.0, keepdim=False):  # pmath.py:128-129
    return 2 / (1 - _c(c, x) * x.pow(2).sum(-1, keepdim=keepdim))


def mobius_add(x, y, c=1.0):  # pmath.py:171-177
    c = _c(c, x)
    x2 = x.pow(2).sum(-1, keepdim=True)
    y2 = y.pow(2).sum(-1, keepdim=True)
    xy = (x * y).sum(-1, keepdim=True)
    num = (1 + 2 * c * xy + c * y2) * x + (1 - c * x2) * y
    den = 1 + 2 * c * xy + c ** 2 * x2 * y2
    return num / (den + 1e-5)


def dist(x, y, c=1.0, keepdim=False):  # pmath.py:205-208
    sc = _c(c, x) ** 0.5
    return artanh(sc * mobius_add(-x, y, c).norm(dim=-1, p=2, keepdim=keepdim)) * 2 / sc


def dist0(x, c=1.0, keepdim=False):  # pmath.py:231-234
    sc = _c(c, x) ** 0.5
    return artanh(sc * x.norm(dim=-1, p=2, keepdim=keepdim)) * 2 / sc


def expmap(x, u, c=1.0):  # pmath.py:268-277
    sc = _c(c, x) ** 0.5
    un = u.norm(dim=-1, p=2, keepdim=True).clamp_min(1e-5)
    second = tanh_clamped(sc / 2 * lambda_x(x, c, keepdim=True) * un) * u / (sc * un)
    return mobius_add(x, second, c)


def expmap0(u, c=1.0):  # pmath.py:300-304
    sc = _c(c, u) ** 0.5
    un = u.norm(dim=-1, p=2, keepdim=True).clamp_min(1e-5)
    return tanh_clamped(sc * un) * u / (sc * un)


def logmap(x, y, c=1.0):  # pmath.py:334-339
    sub = mobius_add(-x, y, c)
    sn = sub.norm(dim=-1, p=2, keepdim=True)
    sc = _c(c, x) ** 0.5
    return 2 / sc / lambda_x(x, c, keepdim=True) * artanh(sc * sn) * sub / sn


def logmap0(y, c=1.0):  # pmath.py:365-368
    sc = _c(c, y) ** 0.5
    yn = y.norm(dim=-1, p=2, keepdim=True).clamp_min(1e-5)
    return y / yn / sc * artanh(sc * yn)


def mobius_matvec(m, x, c=1.0):  # pmath.py:399-408
    sc = _c(c, x) ** 0.5
    xn = x.norm(dim=-1, keepdim=True, p=2).clamp_min(1e-5)
    mx = x @ m.transpose(-1, -2)
    mxn = mx.norm(dim=-1, keepdim=True, p=2)
    res = tanh_clamped(mxn / xn * artanh(sc * xn)) * mx / (mxn * sc)
    zero = (mx == 0).all(dim=-1, keepdim=True)
    return project(torch.where(zero, torch.zeros(1, dtype=res.dtype), res), c)


def mobius_addition_batch(x, y, c=1.0):  # pmath.py:416-427 ; x [B,D], y [C,D] -> [B,C,D]
    c = _c(c, x)
    xy = x @ y.t()
    x2 = x.pow(2).sum(-1, keepdim=True)
    y2 = y.pow(2).sum(-1, keepdim=True)
    num = (1 + 2 * c * xy + c * y2.t()).unsqueeze(2) * x.unsqueeze(1) + (1 - c * x2).unsqueeze(2) * y
    den = 1 + 2 * c * xy + c ** 2 * x2 * y2.t()
    return num / (den.unsqueeze(2) + 1e-5)


def hyperbolic_softmax(X, A, P, c=1.0):  # pmath.py:430-437 ; X [B,D], A,P [C,D] -> [B,C]
    c = _c(c, X)
    lam = 2 / (1 - c * P.pow(2).sum(dim=1))
    k = lam * A.norm(dim=1) / torch.sqrt(c)
    ma = mobius_addition_batch(-P, X, c)  # [C,B,D]
    num = 2 * torch.sqrt(c) * (ma * A.unsqueeze(1)).sum(-1)
    den = A.norm(dim=1, keepdim=True) * (1 - c * ma.pow(2).sum(dim=2))
    return (k.unsqueeze(1) * arsinh(num / den)).t()


def p2k(x, c=1.0):  # pmath.py:440-442
    return 2 * x / (1 + _c(c, x) * x.pow(2).sum(-1, keepdim=True))


def k2p(x, c=1.0):  # pmath.py:445-447
    return x / (1 + torch.sqrt(1 - _c(c, x) * x.pow(2).sum(-1, keepdim=True)))


def lorenz_factor(x, c=1.0, keepdim=False):  # pmath.py:450-469
    return 1 / torch.sqrt(1 - _c(c, x) * x.pow(2).sum(-1, keepdim=keepdim))


def poincare_mean(x, c=1.0):  # pmath.py:472-479 with dim=0 ; x [R,D] -> [D]
    xk = p2k(x, c)
    lam = lorenz_factor(xk, c, keepdim=True)
    mean = (lam * xk).sum(0, keepdim=True) / lam.sum(0, keepdim=True)
    return k2p(mean, c).squeeze(0)


def dist_matrix(x, y, c=1.0):  # pmath.py:482-493 ; [P,D],[R,D] -> [P,R]
    sc = _c(c, x) ** 0.5
    return 2 / sc * artanh(sc * mobius_addition_batch(-x, y, c).norm(dim=-1))


# custom autograd functions of the library (backward rules restated; torch CPU)
def artanh_backward(x, grad):  # Artanh.backward, pmath.py:24-27: the CLAMPED input is what was saved
    xc = x.clamp(-1 + 1e-5, 1 - 1e-5)
    return grad / (1 - xc ** 2)


def arsinh_backward(x, grad):  # Arsinh.backward, pmath.py:57-60
    return grad / (1 + x ** 2) ** 0.5


def riemannian_gradient_backward(x, grad, c=1.0):  # RiemannianGradient.backward, pmath.py:39-45
    return grad * (1 - c * x.pow(2).sum(-1, keepdim=True)).pow(2) / 4
