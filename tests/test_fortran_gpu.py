"""Steps driven from FORTRAN.  (1) the dependency-free driver: fortran/drive_double_gyre (amdflang; fortran/mom6x_c_api.F90 + fortran/mom6x_host.F90 +
libmom6x.so, nothing else) reads a case written here, packs the metric block from arrays with MOM6's symmetric-memory
extents, creates the context, initialises the modules, uploads the state once, runs three steps of
step_MOM_dyn_split_RK2 on the resident state and compares all eight prognostic arrays with the committed fixture
tests/golden/rk2_double_gyre_strong_drag_3steps -- bit for bit, in Fortran.  This is the ISO_C_BINDING boundary of
SURVEY.md 8(b) exercised end to end (the Python mirror the other tests use never touches these code paths:
mom6x_upload / mom6x_download with Fortran extents, the bind(C) struct layouts, mom6x_dev_alloc)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from mom6_amd import abi
from tests import cases
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "fortran", "drive_double_gyre")
STATE = ["u", "v", "h", "uh", "vh", "uhtr", "vhtr", "eta_av"]


def _stagger(name):
    """Staggering code (mom6x_upload) of a metric plane from its MOM6 name."""
    if name.endswith("Bu"):
        return 3
    if name.endswith("Cu") or name == "dy_Cu":
        return 1
    if name.endswith("Cv") or name == "dx_Cv":
        return 2
    return 0


def _f_extent(d, a, stagger):
    """The part of a pitched array that a Fortran array with MOM6's symmetric-memory extents holds (C order [k][j][i])."""
    xB, yB = int(stagger in (1, 3)), int(stagger in (2, 3))
    return np.ascontiguousarray(a[..., d.joff - d.halo - yB: d.joff + d.nj + d.halo, d.ioff - d.halo - xB: d.ioff + d.ni + d.halo])


def _fields(s):
    """A bind(C) struct as Fortran unformatted stream I/O transfers it: component by component, no padding."""
    out = b""
    for name, ctype in s._fields_:
        v = getattr(s, name)
        out += struct.pack("<i", v) if ctype is C.c_int else struct.pack("<d", v)
    return out


def test_three_steps_driven_from_fortran(tmp_path):
    if not os.path.exists(DRIVER):
        pytest.fail("fortran/drive_double_gyre is missing: __graft_entry__.build() compiles it with amdflang")
    cfg = H.double_gyre()
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg)
    cont, bt, cor, pgf, rk2 = cases.rk2_params(d, inp["GV"], dict(strong_drag=1))
    gold = H.load_golden("rk2_double_gyre_strong_drag_3steps")
    nsteps, first_direction = 3, 0
    a_u, a_v, h_u, h_v, Ray_u, Ray_v = inp["coefs"][0]
    path = tmp_path / "case.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<10i", 1297042742, d.ni, d.nj, d.nk, d.halo, nsteps, first_direction, abi.G_COUNT, d.reentrant_x, d.reentrant_y))
        f.write(struct.pack("<d", inp["dt"]))
        for s in (inp["GV"], cont, bt, cor, pgf, rk2):
            f.write(_fields(s))
        f.write(np.ascontiguousarray(inp["Rlay"], dtype="<f8").tobytes()); f.write(np.ascontiguousarray(inp["gp"], dtype="<f8").tobytes())
        for m, name in enumerate(abi.METRICS):
            st = _stagger(name)
            f.write(struct.pack("<i", st)); f.write(_f_extent(d, M[m], st).astype("<f8").tobytes())
        for a, st in ((inp["u"], 1), (inp["v"], 2), (inp["h"], 0), (a_u, 1), (a_v, 2), (h_u, 1), (h_v, 2), (Ray_u, 1), (Ray_v, 2), (inp["taux"], 1), (inp["tauy"], 2)):
            f.write(_f_extent(d, a, st).astype("<f8").tobytes())
        for n in STATE:
            f.write(np.ascontiguousarray(gold[n], dtype="<f8").tobytes())
    r = subprocess.run([DRIVER, str(path)], capture_output=True, text=True, timeout=300)
    print(r.stdout); print(r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count(": bit-identical") == 2 * len(STATE) and r.stdout.count("restart: ") == len(STATE)   # + the restarted run


SHIM_DRIVER = os.path.join(ROOT, "tests", "fortran_stubs", "drive_shims")


def _write_shim_case(path, cfg, orc, no_bt_cont=False):
    """The case of tests/fortran_stubs/drive_shims.F90: grid and state in Fortran extents, the MOM_input table, the
    set_viscous_BBL fields, and the oracle's state after SHIM_NSTEPS steps (= the committed fixture, checked here).
    no_bt_cont: the same with USE_BT_CONT_TYPE = False, NONLINEAR_BT_CONTINUITY = True in the table (no fixture: the oracle's run)."""
    gg, d, M = cfg
    so, m, inp, vis = cases.oracle_shim_case(orc, cfg, no_bt_cont=no_bt_cont)
    stg_of = {"u": "u", "uh": "u", "uhtr": "u", "v": "v", "vh": "v", "vhtr": "v"}
    if no_bt_cont:
        gold = {n: so[n][(Ellipsis,) + tuple(H.interior(d, stg_of.get(n, "h")))] for n in STATE}
    else:
        gold = H.load_golden("rk2_double_gyre_shims_4steps")
        for n in STATE:
            H.assert_bitwise(so[n][(Ellipsis,) + tuple(H.interior(d, stg_of.get(n, "h")))], gold[n], "oracle vs committed fixture: " + n)
    params = cases.shim_case_params(inp["dt"], H.golden_tag())
    if no_bt_cont:
        params.update({"USE_BT_CONT_TYPE": "False", "NONLINEAR_BT_CONTINUITY": "True", "BT_THICK_SCHEME": "HYBRID"})
    GV = inp["GV"]
    with open(path, "wb") as f:
        f.write(struct.pack("<10i", 1297042743, d.ni, d.nj, d.nk, d.halo, cases.SHIM_NSTEPS, cases.SHIM_SAVE_AFTER, 0, abi.G_COUNT, len(params)))
        f.write(struct.pack("<d", inp["dt"]))
        f.write(struct.pack("<9d", GV.g_Earth, GV.Rho0, GV.Angstrom_H, GV.H_subroundoff, GV.dZ_subroundoff, GV.H_to_Z, GV.Z_to_H, GV.H_to_RZ, GV.RZ_to_H))
        f.write(np.ascontiguousarray(inp["Rlay"], dtype="<f8").tobytes()); f.write(np.ascontiguousarray(inp["gp"], dtype="<f8").tobytes())
        for k, v in params.items():
            f.write(k.ljust(48).encode()); f.write(v.ljust(64).encode())

        def put(a, st):
            f.write(struct.pack("<i", st)); f.write(_f_extent(d, a, st).astype("<f8").tobytes())
        for mi, name in enumerate(abi.METRICS):
            put(M[mi], _stagger(name))
        put(inp["u"], 1); put(inp["v"], 2); put(inp["h"], 0)
        put(vis[0], 1); put(vis[1], 2); put(vis[2], 1); put(vis[3], 2); put(vis[4], 0)
        put(inp["taux"], 1); put(inp["tauy"], 2)
        for n in STATE:
            f.write(np.ascontiguousarray(gold[n], dtype="<f8").tobytes())
        # what MOM_diagnostics reads through Accel_diag / MIS after the last step (RK2.F90:1512-1534), from the same oracle run
        for n, stg in DIAG:
            f.write(np.ascontiguousarray(m[n][(Ellipsis,) + tuple(H.interior(d, stg))], dtype="<f8").tobytes())
        # the standalone shims MOM_hor_visc and MOM_ALE on the INITIAL state: a tracer, and the oracle's horizontal_viscosity,
        # ALE_regrid (z*, UNIFORM resolution over the deepest column), ALE_remap_tracers, ALE_remap_set_h_vel x 2 + ALE_remap_velocities
        # (REMAPPING_SCHEME = PLM: the case has two layers) with the parameters the driver's table gives hor_visc_init / ALE_init
        from mom6_amd import synth
        hv = abi.hor_visc_params_default(inp["dt"]); hv.Ah_vel_scale = 0.02; hv.Smagorinsky_Ah = 1; hv.Smag_bi_const = 0.06
        u0, v0, h0 = inp["u"], inp["v"], inp["h"]
        du, dv = np.zeros_like(u0), np.zeros_like(v0)
        orc.horizontal_viscosity(d, M, GV, hv, orc.hor_visc_init(d, M, hv), u0, v0, h0, du, dv)
        max_depth = float(M[abi.G["bathyT"]].max())
        cr = np.full(d.nk, max_depth / d.nk)
        tot = 0.0
        for x in cr:
            tot = tot + x
        cr[-1] = cr[-1] + (max_depth - tot)                      # initialize_regridding MOM_regridding.F90:563-582
        RP = abi.regrid_zstar_params_default()
        CS = abi.remapping_params_default(abi.REMAP_PLM, GV.H_subroundoff, om4_remap_via_sub_cells=1, boundary_extrapolation=0)
        hn, dz = np.zeros_like(h0), np.zeros((d.nk + 1,) + d.shape2())
        orc.ALE_regrid_zstar(d, M, GV, RP, cr, h0, hn, dz)
        T0 = np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 70, nk=d.nk, ox=0.5, oy=0.5) + np.arange(d.nk)[:, None, None])
        Tr = T0.copy()
        orc.ALE_remap_tracers(d, M, CS, h0, hn, [Tr])
        huo, hvo, hun, hvn = (np.zeros_like(h0) for _ in range(4))
        orc.ALE_remap_set_h_vel(d, M, h0, huo, hvo); orc.ALE_remap_set_h_vel(d, M, hn, hun, hvn)
        ur, vr = u0.copy(), v0.copy()
        orc.ALE_remap_velocities(d, M, CS, huo, hvo, hun, hvn, ur, vr)
        assert np.abs(du).max() > 0 and np.abs(Tr - T0).max() > 0 and np.abs(ur - u0).max() > 0
        f.write(struct.pack("<i", 1297042744)); f.write(struct.pack("<d", max_depth))
        put(T0, 0)
        for a, stg in ((du, "u"), (dv, "v"), (hn, "h"), (dz, "h"), (Tr, "h"), (ur, "u"), (vr, "v")):
            f.write(np.ascontiguousarray(a[(Ellipsis,) + tuple(H.interior(d, stg))], dtype="<f8").tobytes())
        # REGRIDDING_COORDINATE_MODE = RHO through MOM_ALE: the linear equation of state of the table's defaults, target densities as
        # set_target_densities_from_GV (MOM_regridding.F90:2069) makes them from GV%Rlay with the UNIFORM resolution, T and S around
        # the layer densities with enough noise for statically unstable columns; convective_adjustment, then the new grid
        eos = abi.eos_params_default(abi.LINEAR)
        R = np.asarray(inp["Rlay"], dtype=np.float64); ke = d.nk
        rho_light = R[0] + 0.5 * (R[0] - R[min(1, ke - 1)]); rho_heavy = R[ke - 1] + 0.5 * (R[ke - 1] - R[max(ke - 2, 0)])
        res = np.full(ke, (rho_heavy - rho_light) / ke)
        tgt = np.zeros(ke + 1)
        tgt[0] = R[0] + 0.5 * (R[0] - R[1]); tgt[ke] = R[ke - 1] + 0.5 * (R[ke - 1] - R[ke - 2])
        for k in range(1, ke):
            tgt[k] = tgt[k - 1] + res[k]
        rng = np.random.default_rng(11)
        Srho = np.full_like(h0, 35.0)
        Trho = np.ascontiguousarray((1000.0 + 0.8 * 35.0 - R)[:, None, None] / 0.2 + 3.0 * synth.smooth_field(d, 71, nk=ke) + rng.normal(0.0, 4.0, h0.shape))
        RPr = abi.regrid_rho_params_default(integrate_downward_for_e=0)      # ALE_init: REGRID_USE_OLD_DIRECTION = True (MOM_ALE.F90:314-319)
        hy, Ty, Sy = h0.copy(), Trho.copy(), Srho.copy()
        orc.ALE_convective_adjustment(d, eos, hy, Ty, Sy)
        hny, dzy = np.zeros_like(h0), np.zeros((ke + 1,) + d.shape2())
        orc.ALE_regrid_rho(d, M, GV, RPr, eos, tgt, hy, Ty, Sy, hny, dzy)
        assert np.abs(Ty - Trho).max() > 0 and np.abs(dzy).max() > 0
        put(Trho, 0); put(Srho, 0)
        for a in (hy, Ty, Sy, hny, dzy):
            f.write(np.ascontiguousarray(a[(Ellipsis,) + tuple(H.interior(d, "h"))], dtype="<f8").tobytes())


N_STANDALONE = 15     # horizontal_viscosity (2), ALE_regrid (2), ALE_remap_tracers (1), ALE_remap_velocities (2), the RHO coordinate's
                      # pre_ALE_adjustments (3) and ALE_regrid (2) through MOM_hor_visc / MOM_ALE, the remapping again on resident arrays (3)
DIAG = [("CAu", "u"), ("CAv", "v"), ("PFu", "u"), ("PFv", "v"), ("diffu", "u"), ("diffv", "v"), ("u_accel_bt", "u"), ("v_accel_bt", "v"),
        ("pbce", "h"), ("u_av", "u"), ("v_av", "v")]


@pytest.mark.parametrize("mode", ["", "resident"])
def test_shim_modules_new_run_restart_and_tracers_from_fortran(orc, tmp_path, mode, sums):
    """The shim MODULES (fortran/shims/: MOM_dynamics_split_RK2 -> MOM_continuity_PPM, MOM_barotropic, MOM_CoriolisAdv,
    MOM_PressureForce, MOM_vert_friction; MOM_tracer_advect; mom6x_diabatic_solvers), compiled against the interface stand-ins
    of tests/fortran_stubs/, driven the way MOM.F90 drives them: four steps uninterrupted == the oracle's fixture bit for bit;
    two steps, save_restart (every registered variable, the barotropic ones included), end, a fresh control structure,
    restore_state, two more steps == the same fixture bit for bit.  Both modes of the shim: state across PCIe every step
    (MOM.F90 unchanged) and resident in HBM (three hook calls).  Every arithmetic of the column sums.  The driver also calls the
    sub-module shims (CorAdCalc, btcalc, bt_mass_source, btstep) on plain host arrays and on arrays made resident with
    mom6x_shim_ctx's shim_resident_add, counting the arrays that cross PCIe (29 against 0) and comparing the results."""
    if not os.path.exists(SHIM_DRIVER):
        pytest.fail("tests/fortran_stubs/drive_shims is missing: __graft_entry__.build() compiles it with amdflang")
    path = tmp_path / "shim_case.bin"
    _write_shim_case(path, H.double_gyre(), orc)
    r = subprocess.run([SHIM_DRIVER, str(path), str(tmp_path / "restart.bin")] + ([mode] if mode else []),
                       capture_output=True, text=True, timeout=300)
    print(r.stdout); print(r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    # runs A and C: the state; run C also: the arrays behind Accel_diag% and MIS% (associated, and equal to the oracle's)
    assert r.stdout.count(": bit-identical") == 2 * len(STATE) + len(DIAG) + 7 + N_STANDALONE
    # CorAdCalc, btcalc, bt_mass_source and btstep through their shims on arrays the host has made resident (shim_resident_add): not one
    # array crosses PCIe in two rounds of calls, and the seven results equal those of the calls on plain host arrays
    assert ", resident 0" in r.stdout and r.stdout.count("resident CorAdCalc") == 2 and r.stdout.count("resident btstep") == 5
    assert "ALE_remap_tracers + 2 x ALE_remap_set_h_vel + ALE_remap_velocities: 0" in r.stdout
    assert "D (restart file without CAu, CAv) u: max |diff|" in r.stdout
    names = r.stdout.split("registered restart variables:")[1].splitlines()[0].split()
    assert names == ["u", "v", "h", "sfc", "u2", "v2", "CAu", "CAv", "diffu", "diffv", "ubtav", "vbtav", "DTBT"]
    _check_checksum_lines(r.stdout, H.double_gyre(), orc)


def _check_checksum_lines(stdout, cfg, orc):
    """The lines hchksum / uvchksum / Bchksum / hchksum_pair wrote through fortran/shims/MOM_checksums.F90 == the lines the Python host
    formats from the same device routine, whose numbers equal the oracle's."""
    import torch
    from mom6_amd import synth
    from mom6_amd.dycore import Dycore
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg, False, False)
    dyc = Dycore(d, M, inp["GV"])
    T0 = np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 70, nk=d.nk, ox=0.5, oy=0.5) + np.arange(d.nk)[:, None, None])
    f = np.ascontiguousarray(M[abi.G["CoriolisBu"]])
    want = []
    for arr, stg, mesg, kw in ((inp["h"], "h", "h0 [MOM_checksums]", dict(haloshift=1)),
                               (inp["u"], "u", "u uv0 [MOM_checksums]", dict(haloshift=1, symmetric=True)),
                               (inp["v"], "v", "v uv0 [MOM_checksums]", dict(haloshift=1, symmetric=True)),
                               (f, "B", "f [MOM_checksums]", dict(haloshift=0, symmetric=True)),
                               (inp["h"], "h", "h0 x2 [MOM_checksums]", dict(scale=2.0)),
                               (inp["h"], "h", "x hT [MOM_checksums]", dict(haloshift=2, omit_corners=True)),
                               (T0, "h", "y hT [MOM_checksums]", dict(haloshift=2, omit_corners=True))):
        a = dyc.to_dev(arr)
        got = dyc.chksum(a, stg, **kw); got.pop("kind")
        assert got == orc.chksum(d, arr, stg, **kw), (mesg, got)
        want += [l.rstrip() for l in dyc.chksum_lines(a, stg, mesg, **kw)]
    have = [l.rstrip() for l in stdout.splitlines() if "[MOM_checksums]" in l]
    assert have == want, "\n".join(["Fortran:"] + have + ["Python:"] + want)


def test_shim_modules_without_a_BT_cont_type_from_fortran(orc, tmp_path, sums):
    """The same driver with USE_BT_CONT_TYPE = False and NONLINEAR_BT_CONTINUITY = True in its MOM_input table : MOM_barotropic's barotropic_init reads them, initialize_dyn_split_RK2 passes no_BT_cont on, and four
    steps uninterrupted as well as two + restart + two equal the oracle's run of that configuration bit for bit."""
    if not os.path.exists(SHIM_DRIVER):
        pytest.fail("tests/fortran_stubs/drive_shims is missing: __graft_entry__.build() compiles it with amdflang")
    path = tmp_path / "shim_case.bin"
    _write_shim_case(path, H.double_gyre(), orc, no_bt_cont=True)
    r = subprocess.run([SHIM_DRIVER, str(path), str(tmp_path / "restart.bin")], capture_output=True, text=True, timeout=300)
    print(r.stdout); print(r.stderr)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count(": bit-identical") == 2 * len(STATE) + len(DIAG) + 7 + N_STANDALONE
