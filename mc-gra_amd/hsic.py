"""hsic.py of the reference (/root/reference/MC-GRA/hsic.py) on the MI355X: same function names and arguments, device
tensors in, device scalars out.  The kernel matrices, their centring and every mean run in libmcgra_hip.so
(csrc/capi.hip); what the reference itself does on the host stays on the host: ``sigma_estimation`` takes the median
of the pairwise distances with numpy (hsic.py:5-17) after the distance matrix has been formed on the device.

``hsic_normalized_cca`` (:138-151) is evaluated in fp64 on the device (its two regularised m x m inverses are
ill-conditioned: the reference's fp32 result carries up to 1e-2 of error, see csrc/capi.hip).
``use_cuda`` / ``to_numpy`` are accepted and ignored (the reference ignores ``to_numpy`` too).
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .engine import _on_operand_device, _p, _stream


def _f32(x):
    return x.detach().to(dtype=torch.float32).contiguous()


@_on_operand_device
def distmat(X):
    """hsic.distmat (:20-27)."""
    X = _f32(X)
    out = torch.empty(X.shape[0], X.shape[0], device=X.device, dtype=torch.float32)
    check(lib.mcgra_distmat(_stream(), X.shape[0], X.shape[1], _p(X), _p(out)))
    return out


def sigma_estimation(X, Y):
    """hsic.sigma_estimation (:5-17): median of the pairwise squared distances of cat([X, Y]) (lower triangle)."""
    D = distmat(torch.cat([_f32(X), _f32(Y)])).cpu().numpy()
    tri = D[np.tril_indices(D.shape[0], -1)]
    med = np.median(tri)
    if med <= 0:
        med = np.mean(tri)
    if med < 1E-2:
        med = 1E-2
    return float(med)


def distcorr(X, sigma=1.0):
    """hsic.distcorr (:50-53): mean(exp(-distmat(X) / (2 sigma^2)))."""
    return _gauss_mean(_f32(X), float(sigma))


def _gauss_mean(X, sigma):
    """mean(Kx) through mcgra_mmd against a single point with an enormous bandwidth: Ky = Kxy = 1, so mmd = mean(Kx) - 1."""
    out = torch.zeros(1, device=X.device, dtype=torch.float32)
    big = 1e18
    y = torch.zeros(1, X.shape[1], device=X.device, dtype=torch.float32)
    with torch.cuda.device(X.device):
        check(lib.mcgra_mmd(_stream(), X.shape[0], 1, X.shape[1], _p(X), _p(y), float(sigma), big, big, _p(out)))
    return out[0] + 1.0


@_on_operand_device
def mmd(x, y, sigma=None, use_cuda=True, to_numpy=False):
    """hsic.mmd (:68-89)."""
    x, y = _f32(x), _f32(y)
    if sigma:
        sx = sy = sxy = float(sigma)
    else:
        sx, sy, sxy = sigma_estimation(x, x), sigma_estimation(y, y), sigma_estimation(x, y)
    out = torch.zeros(1, device=x.device, dtype=torch.float32)
    check(lib.mcgra_mmd(_stream(), x.shape[0], y.shape[0], x.shape[1], _p(x), _p(y), sx, sy, sxy, _p(out)))
    return out[0]


@_on_operand_device
def mmd_pxpy_pxy(x, y, sigma=None, use_cuda=True, to_numpy=False):
    """hsic.mmd_pxpy_pxy (:92-114)."""
    x, y = _f32(x), _f32(y)
    if sigma:
        sx = sy = float(sigma)
    else:
        sx, sy = sigma_estimation(x, x), sigma_estimation(y, y)
    out = torch.zeros(1, device=x.device, dtype=torch.float32)
    check(lib.mcgra_mmd_pxpy_pxy(_stream(), x.shape[0], x.shape[1], y.shape[1], _p(x), _p(y), sx, sy, _p(out)))
    return out[0]


def _hsic(x, y, sigma, normalized):
    x, y = _f32(x), _f32(y)
    if sigma:
        sx = sy = float(sigma)
    else:                     # kernelmat (:39-41): one estimate per operand
        sx, sy = sigma_estimation(x, x), sigma_estimation(y, y)
    out = torch.zeros(1, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        check(lib.mcgra_hsic_regular2(_stream(), x.shape[0], x.shape[1], y.shape[1], _p(x), _p(y), sx, sy, int(normalized), _p(out)))
    return out[0]


def hsic_regular(x, y, sigma=None, use_cuda=True, to_numpy=False):
    """hsic.hsic_regular (:117-124)."""
    return _hsic(x, y, sigma, False)


def hsic_normalized(x, y, sigma=None, use_cuda=True, to_numpy=True):
    """hsic.hsic_normalized (:127-135)."""
    return _hsic(x, y, sigma, True)


@_on_operand_device
def hsic_normalized_cca(x, y, sigma=None, use_cuda=True, to_numpy=True):
    """hsic.hsic_normalized_cca (:138-151); utils.hsic_normalized_cca (utils.py:732-743) is this with sigma=5.0."""
    x, y = _f32(x), _f32(y)
    if sigma:
        sx = sy = float(sigma)
    else:
        sx, sy = sigma_estimation(x, x), sigma_estimation(y, y)
    out = torch.zeros(1, device=x.device, dtype=torch.float32)
    check(lib.mcgra_hsic_normalized_cca(_stream(), x.shape[0], x.shape[1], y.shape[1], _p(x), _p(y), sx, sy, _p(out)))
    return out[0]
