"""horizontal_viscosity / hor_visc_init on the device against the oracle (MOM_hor_visc.F90) -- bit for bit."""
import numpy as np
import pytest

from mom6_amd import abi, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

FLAGS = {
    "default_biharmonic": dict(Ah_vel_scale=0.01),
    "laplacian_only": dict(Laplacian=1, biharmonic=0, Kh=500.0, Kh_vel_scale=0.02),
    "both_better_bounds": dict(Laplacian=1, Kh=2.0e3, Kh_vel_scale=0.05, Ah=1.0e11, Ah_vel_scale=0.05, Ah_time_scale=8.0e4),
    "smagorinsky": dict(Laplacian=1, Smagorinsky_Kh=1, Smag_Lap_const=0.15, Smagorinsky_Ah=1, Smag_bi_const=0.06, Kh=10.0,
                        Ah=1.0e8),
    # the OM4-class switch set, which k_hv_fused has a compile-time instantiation for (and with coefficients large enough for
    # the "better" bounds and the backscatter branch to bind)
    "om4_class": dict(Laplacian=1, Kh_vel_scale=0.01, Ah_vel_scale=0.01, Smagorinsky_Ah=1, Smag_bi_const=0.06),
    "om4_class_bounds_binding": dict(Laplacian=1, Kh=2.0e4, Ah=1.0e13, Smagorinsky_Ah=1, Smag_bi_const=0.5),
    "smagorinsky_bound_coriolis": dict(Smagorinsky_Ah=1, Smag_bi_const=0.06, bound_Coriolis=1, bound_Cor_vel=2.0, Ah=1.0e8),
    "les_added_legacy_bounds": dict(Laplacian=1, Smagorinsky_Kh=1, Smag_Lap_const=0.15, add_LES_viscosity=1, Kh=50.0,
                                    better_bound_Kh=0, better_bound_Ah=0, Smagorinsky_Ah=1, Smag_bi_const=0.06,
                                    Kh_bg_min=20.0),
    "unbounded": dict(Laplacian=1, Kh=300.0, Ah=5.0e9, bound_Kh=0, bound_Ah=0, better_bound_Kh=0, better_bound_Ah=0),
    "better_Ah_only": dict(Laplacian=1, Kh=300.0, Ah=5.0e12, better_bound_Kh=0, bound_Kh=0),
    "noslip_laplacian": dict(Laplacian=1, biharmonic=0, Kh=800.0, no_slip=1),
    "no_land_mask_no_backscatter": dict(Laplacian=1, Kh=2.0e4, Ah=1.0e13, use_land_mask=0, backscatter_underbound=0),
    # Leith (1996) viscosities from the vorticity gradient: LEITH_KH with its default USE_BETA_IN_LEITH, LEITH_AH,
    # MODIFIED_LEITH (+ the divergence gradient), ADD_LES_VISCOSITY / legacy bounds, with Smagorinsky next to them, NOSLIP
    "leith_kh_beta": dict(Laplacian=1, Leith_Kh=1, Leith_Lap_const=1.0, use_beta_in_Leith=1, Kh=10.0, Ah=1.0e8),
    "leith_kh_ah_modified": dict(Laplacian=1, Leith_Kh=1, Leith_Lap_const=1.5, Leith_Ah=1, Leith_bi_const=2.0, modified_Leith=1,
                                 use_beta_in_Leith=1, Kh=10.0, Ah=1.0e8),
    "leith_ah_only": dict(Leith_Ah=1, Leith_bi_const=5.0, Ah=1.0e7),
    "leith_les_added_legacy_bounds_smag": dict(Laplacian=1, Leith_Kh=1, Leith_Lap_const=1.0, add_LES_viscosity=1, Smagorinsky_Kh=1,
                                               Smag_Lap_const=0.15, better_bound_Kh=0, better_bound_Ah=0, Leith_Ah=1,
                                               Leith_bi_const=1.0, Smagorinsky_Ah=1, Smag_bi_const=0.06, modified_Leith=1),
    "leith_noslip_laplacian": dict(Laplacian=1, biharmonic=0, Leith_Kh=1, Leith_Lap_const=2.0, use_beta_in_Leith=1, no_slip=1,
                                   Kh=100.0),
}


def hv_params(mods, dt=1200.0):
    P = abi.hor_visc_params_default(dt)
    for k, v in mods.items():
        setattr(P, k, v)
    return P


@pytest.mark.parametrize("cfg", ["island_basin", "benchmark_small", "channel"])
@pytest.mark.parametrize("flags", sorted(FLAGS))
def test_horizontal_viscosity(orc, cfg, flags):
    import torch
    from mom6_amd.dycore import Dycore
    gg, d, M = getattr(H, cfg)()
    if cfg == "island_basin":
        M = H.partial_faces(d, M)
    GV = abi.vgrid_default()
    P = hv_params(FLAGS[flags])
    h, u, v = synth.make_state(d, M, thin_frac=0.15)
    planes = orc.hor_visc_init(d, M, P)
    o_du, o_dv = np.zeros_like(u), np.zeros_like(v)
    orc.horizontal_viscosity(d, M, GV, P, planes, u, v, h, o_du, o_dv)
    dyc = Dycore(d, M, GV)
    dyc.hor_visc_init(P)
    ud, vd, hd = dyc.to_dev(u), dyc.to_dev(v), dyc.to_dev(h)
    du, dv = torch.zeros_like(ud), torch.zeros_like(vd)
    torch.cuda.synchronize()
    dyc.horizontal_viscosity(ud, vd, hd, du, dv)
    dyc.sync()
    H.assert_bitwise(du.cpu().numpy(), o_du, "diffu", H.interior(d, "u"))
    H.assert_bitwise(dv.cpu().numpy(), o_dv, "diffv", H.interior(d, "v"))
    assert np.isfinite(o_du).all() and np.abs(o_du).max() > 0 and np.abs(o_dv).max() > 0
    if "leith" in flags:   # the Leith terms are felt: the same call without them gives another answer
        P0 = hv_params({k: v for k, v in FLAGS[flags].items() if "Leith" not in k and "leith" not in k.lower()})
        r_du, r_dv = np.zeros_like(u), np.zeros_like(v)
        orc.horizontal_viscosity(d, M, GV, P0, orc.hor_visc_init(d, M, P0), u, v, h, r_du, r_dv)
        assert not np.array_equal(r_du, o_du)
    dyc.close()


def test_hor_visc_init_rejects_noslip_biharmonic():
    """hor_visc_init :2723-2725 is a FATAL in the reference."""
    from mom6_amd.dycore import Dycore
    gg, d, M = H.double_gyre()
    dyc = Dycore(d, M, abi.vgrid_default())
    with pytest.raises(RuntimeError, match="NOSLIP and BIHARMONIC"):
        dyc.hor_visc_init(hv_params(dict(no_slip=1)))
    with pytest.raises(RuntimeError, match="initialized"):
        import torch
        z = torch.zeros(d.nk, *d.shape2(), dtype=torch.float64, device="cuda")
        dyc.horizontal_viscosity(z, z, z, z.clone(), z.clone())
    dyc.close()


def test_the_other_variants_are_bit_identical_too():
    """The default is k_hv_fused on 32 x 24 tiles (the four stages in one LDS-tiled kernel, hor_visc.hip).  The four-kernel chain
    (MOM6X_HORVISC=legacy; what Leith configurations always take) is held to the same oracle: this file again in a process with the
    switch set.  (The generic instantiation of k_hv_fused is what every case but the OM4-class one above runs.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("legacy",):
        env = dict(os.environ, MOM6X_HORVISC=mode)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_horvisc_gpu.py"), "-m", "gpu", "-q", "-x",
                            "-k", "not other_variants"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
