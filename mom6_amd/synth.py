"""Seeded synthetic model states on a tile (host numpy, pitched layout).

SURVEY.md section 8d: smooth seeded fields -- layer thicknesses = resting thickness
perturbed by <= a few %, velocities from a smooth field with |u| <= u_max, masked by
the grid masks.  Every field is generated from GLOBAL index coordinates so that a
multi-tile decomposition sees exactly the same global state as a single tile.
"""
import numpy as np

from .abi import G


def _coords(d):
    nrows, P = d.nj + 2 * d.halo + 1, d.pitch
    il = np.arange(P) - d.ioff + d.i_glob0
    jl = np.arange(nrows) - d.joff + d.j_glob0
    if d.reentrant_x:
        il = np.mod(il, d.ni_glob)
    if d.reentrant_y:
        jl = np.mod(jl, d.nj_glob)
    return il[None, :].astype(np.float64), jl[:, None].astype(np.float64)


def smooth_field(d, seed, nk=None, nmodes=6, ox=0.0, oy=0.0):
    """Sum of a few random global Fourier modes, values in about [-1, 1].

    (ox, oy) = staggering offset of the point within the cell (0.5,0.5 = T point)."""
    rng = np.random.default_rng(seed)
    x, y = _coords(d)
    x = (x + ox) / d.ni_glob
    y = (y + oy) / d.nj_glob
    nkk = 1 if nk is None else nk
    out = np.zeros((nkk, x.shape[1] and y.shape[0], x.shape[1]))
    for k in range(nkk):
        acc = np.zeros((y.shape[0], x.shape[1]))
        for _ in range(nmodes):
            kx, ky = rng.integers(1, 5), rng.integers(1, 5)
            ph1, ph2 = rng.uniform(0, 2 * np.pi, 2)
            amp = rng.uniform(0.3, 1.0)
            acc += amp * np.sin(2 * np.pi * kx * x + ph1) * np.sin(2 * np.pi * ky * y + ph2)
        out[k] = acc / (0.65 * nmodes)
    return out if nk is not None else out[0]


def make_state(d, M, seed=20250808, u_max=0.5, h_pert=0.02, thin_frac=0.0):
    """h [nk], u [nk], v [nk] : layered state consistent with the bathymetry of metrics M.

    Column thickness = bathyT (so eta ~ 0 + perturbation); each layer gets 1/nk of it times
    (1 + h_pert*smooth).  With thin_frac>0 some layers are made nearly vanished (Angstrom-like),
    exercising the positive-definite limiter."""
    nk = d.nk
    bathy = M[G["bathyT"]]
    h = np.zeros(d.shape3())
    pert = smooth_field(d, seed, nk=nk, ox=0.5, oy=0.5)
    for k in range(nk):
        h[k] = (bathy / nk) * (1.0 + h_pert * pert[k])
    if thin_frac > 0.0:
        thin = smooth_field(d, seed + 7, nk=nk, ox=0.5, oy=0.5) > (1.0 - 2.0 * thin_frac)
        h = np.where(thin, 1e-9 + 1e-3 * np.abs(pert), h)
    h = np.where(M[G["mask2dT"]][None] > 0, np.maximum(h, 1e-10), 1e-10 * np.ones_like(h))
    u = u_max * smooth_field(d, seed + 1, nk=nk, ox=1.0, oy=0.5) * M[G["mask2dCu"]][None]
    v = u_max * smooth_field(d, seed + 2, nk=nk, ox=0.5, oy=1.0) * M[G["mask2dCv"]][None]
    return np.ascontiguousarray(h), np.ascontiguousarray(u), np.ascontiguousarray(v)
