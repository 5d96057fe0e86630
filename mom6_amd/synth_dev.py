"""Synthetic model state generated directly in HBM (torch), for full-size benchmarks where host
generation of dozens of ~1 GB arrays would dominate.  Same recipe as synth.make_state (smooth global
Fourier modes, masked), not bit-identical to it."""
import math

import numpy as np
import torch

from .abi import G


def _coords(d, device, ox, oy):
    nrows, P = d.nj + 2 * d.halo + 1, d.pitch
    il = torch.arange(P, device=device, dtype=torch.float64) - d.ioff + d.i_glob0
    jl = torch.arange(nrows, device=device, dtype=torch.float64) - d.joff + d.j_glob0
    if d.reentrant_x:
        il = torch.remainder(il, d.ni_glob)
    if d.reentrant_y:
        jl = torch.remainder(jl, d.nj_glob)
    return ((il + ox) / d.ni_glob)[None, :], ((jl + oy) / d.nj_glob)[:, None]


def smooth_field(d, device, seed, nk=None, ox=0.0, oy=0.0, nmodes=4):
    rng = np.random.default_rng(seed)
    x, y = _coords(d, device, ox, oy)
    nkk = 1 if nk is None else nk
    out = torch.empty((nkk, y.shape[0], x.shape[1]), dtype=torch.float64, device=device)
    for k in range(nkk):
        acc = torch.zeros((y.shape[0], x.shape[1]), dtype=torch.float64, device=device)
        for _ in range(nmodes):
            kx, ky = int(rng.integers(1, 5)), int(rng.integers(1, 5))
            ph1, ph2 = rng.uniform(0, 2 * math.pi, 2)
            amp = rng.uniform(0.3, 1.0)
            acc += amp * torch.sin(2 * math.pi * kx * x + ph1) * torch.sin(2 * math.pi * ky * y + ph2)
        out[k] = acc / (0.65 * nmodes)
    return out if nk is not None else out[0]


def make_state(d, Mdev, seed=20250808, u_max=0.5, h_pert=0.02):
    dev = Mdev.device
    nk = d.nk
    bathy = Mdev[G["bathyT"]]
    pert = smooth_field(d, dev, seed, nk=nk, ox=0.5, oy=0.5)
    h = (bathy / nk)[None] * (1.0 + h_pert * pert)
    h = torch.where(Mdev[G["mask2dT"]][None] > 0, torch.clamp(h, min=1e-10), torch.full_like(h, 1e-10))
    del pert
    u = u_max * smooth_field(d, dev, seed + 1, nk=nk, ox=1.0, oy=0.5) * Mdev[G["mask2dCu"]][None]
    v = u_max * smooth_field(d, dev, seed + 2, nk=nk, ox=0.5, oy=1.0) * Mdev[G["mask2dCv"]][None]
    return h.contiguous(), u.contiguous(), v.contiguous()


def make_state_coherent(d, Mdev, seed=20250808, u_max=0.5, h_pert=0.02):
    """A second, LABELLED state for the mass-flux kernels' direction statistics (profiles/r05_mfw.md): the same thicknesses, but
    velocities that are vertically coherent as an ocean column's are -- a barotropic current plus two baroclinic modes (a
    surface-intensified first mode and a second mode with one zero crossing more) with smooth horizontal amplitudes -- instead of
    make_state's independent random current in every layer.  The benchmark's headline stays on make_state."""
    dev = Mdev.device
    nk = d.nk
    h, _, _ = make_state(d, Mdev, seed=seed, u_max=0.0, h_pert=h_pert)
    z = (torch.arange(nk, device=dev, dtype=torch.float64) + 0.5) / nk             # 0 at the surface, 1 at the bottom
    m1 = torch.exp(-z / 0.15); m1 = m1 - m1.mean()                                  # surface-intensified
    m2 = torch.cos(2.0 * math.pi * z) * torch.exp(-z / 0.5); m2 = m2 - m2.mean()
    m1 = m1 / m1.abs().max(); m2 = m2 / m2.abs().max()

    def vel(s0, ox, oy, mask):
        bt = smooth_field(d, dev, s0, ox=ox, oy=oy)
        a1 = smooth_field(d, dev, s0 + 10, ox=ox, oy=oy)
        a2 = smooth_field(d, dev, s0 + 20, ox=ox, oy=oy)
        f = 0.5 * bt[None] + 0.35 * a1[None] * m1[:, None, None] + 0.15 * a2[None] * m2[:, None, None]
        return (u_max * f * mask[None]).contiguous()

    u = vel(seed + 1, 1.0, 0.5, Mdev[G["mask2dCu"]])
    v = vel(seed + 2, 0.5, 1.0, Mdev[G["mask2dCv"]])
    return h.contiguous(), u, v
