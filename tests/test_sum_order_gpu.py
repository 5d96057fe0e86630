"""The device's DEFAULT order of the mass-flux column sums (mom6x_continuity_params.sum_order = MOM6X_SUM_TREE16, the wave-owned
kernel of continuity_wave.hip) against the oracle run in the REFERENCE's order (sequential in k: zonal/meridional_flux_adjust
MOM_continuity_PPM.F90:1093-1242, set_zonal/merid_BT_cont :1246-1409 with the duL / duR recurrences :1293-1316).

The two orders are different floating-point programs, so this is a TOLERANCE test, the only one on the path: the whole
split-RK2 step with every callee on the device (vertvisc_coef, horizontal_viscosity), device and oracle stepped side by side,
the difference of every prognostic and restart field measured after every step as a fraction of the field's range.
Stated bound: 1e-11 of range per field after 10 steps at nk = 75 (BOUND below; the observed numbers are printed and, on the
GPU box, written to gpurun_out/sum_order_drift.json -- README.md / DESIGN.md section 2 quote them).  Independent of the
order: sum_k uh = uhbt to ETA_TOLERANCE (the Newton solve's own stopping rule :1178) and the basin's volume to round-off.

The same file holds the opposite check for the exact mode: device REFERENCE order vs oracle REFERENCE order over the same 10
steps is bit-identical (zeros of opposite sign not tolerated)."""
import json
import os

import numpy as np
import pytest

from mom6_amd import abi
from tests import cases
from tests import helpers as H
from tests.test_rk2_gpu import STAG, STATE

pytestmark = pytest.mark.gpu
G = abi.G

BOUND = 1.0e-11           # of max |field|, after the last step; observed (gpurun_out/sum_order_drift.json, copied to profiles/): <= 3.3e-12
AUX = ["CAu_pred", "CAv_pred", "diffu", "diffv", "visc_rem_u", "visc_rem_v", "u_av", "v_av", "h_av", "eta", "uhbt", "vhbt"]


def _pair(orc, cfg, dev_order, orc_order, strong_drag=1):
    """A device model and an oracle model of the same seeded case, each with its own order of the column sums."""
    import torch
    from mom6_amd.dycore import Dycore
    from tests.test_dyn_gpu import visc_inputs
    gg, d, M = cfg
    inp = cases.rk2_inputs(cfg, False, False)
    GV, Rlay, gp, dt = inp["GV"], inp["Rlay"], inp["gp"], inp["dt"]
    bt_mod = dict(strong_drag=strong_drag)
    P = abi.vertvisc_params_default()
    vvset = (P,) + tuple(visc_inputs(d, M)) + (inp["coefs"][0][4], inp["coefs"][0][5])
    hv = abi.hor_visc_params_default(dt, Laplacian=True, biharmonic=True)
    hv.Kh_vel_scale = 0.01; hv.Ah_vel_scale = 0.01; hv.Smagorinsky_Ah = 1; hv.Smag_bi_const = 0.06
    # oracle
    cont, bt, cor, pgf, rk2 = cases.rk2_params(d, GV, bt_mod, None, None)
    cont.sum_order = orc_order
    m = orc.OrcModel(d, M, GV, cont, bt, cor, pgf, rk2, Rlay, gp, 0)
    m.set_vertvisc(*vvset); m.set_hor_visc(hv)
    so = dict(u=inp["u"].copy(), v=inp["v"].copy(), h=inp["h"].copy(), uh=np.zeros_like(inp["h"]), vh=np.zeros_like(inp["h"]),
              uhtr=np.zeros_like(inp["h"]), vhtr=np.zeros_like(inp["h"]), eta_av=np.zeros(d.shape2()))
    m.initialize(so["u"], so["v"], so["h"], so["uh"], so["vh"], dt)
    # device
    cont2, bt2, cor2, pgf2, rk22 = cases.rk2_params(d, GV, bt_mod, None, None)
    cont2.sum_order = dev_order
    dyc = Dycore(d, M, GV, 0)
    dyc.continuity_init(cont2); dyc.barotropic_init(bt2); dyc.CoriolisAdv_init(cor2); dyc.PressureForce_init(pgf2, Rlay, gp)
    dyc.initialize_dyn_split_RK2(rk22)
    sg = dict(u=dyc.to_dev(inp["u"]), v=dyc.to_dev(inp["v"]), h=dyc.to_dev(inp["h"]), uh=dyc.zeros3(), vh=dyc.zeros3(),
              uhtr=dyc.zeros3(), vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
    dyc.vertvisc_init(vvset[0])
    keep = [dyc.to_dev(a) if a is not None else None for a in vvset[1:]]
    dyc.vertvisc_set_visc(*keep)
    dyc.hor_visc_init(hv)
    tx, ty = dyc.to_dev(inp["taux"]), dyc.to_dev(inp["tauy"])
    torch.cuda.synchronize()
    dyc.dyn_split_RK2_new_run(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], dt)

    def step(n):
        m.step(so["u"], so["v"], so["h"], so["uh"], so["vh"], so["uhtr"], so["vhtr"], so["eta_av"], inp["taux"], inp["tauy"], dt,
               inp["coefs"], calc_dtbt=(n == 0))
        dyc.step_MOM_dyn_split_RK2(sg["u"], sg["v"], sg["h"], sg["uh"], sg["vh"], sg["uhtr"], sg["vhtr"], sg["eta_av"], tx, ty,
                                   dt, calc_dtbt=(n == 0))
        dyc.sync()
    return dict(d=d, M=M, dt=dt, inp=inp, cont=cont2, m=m, so=so, dyc=dyc, sg=sg, step=step, keep=(keep, tx, ty))


def _fields(p):
    dyc, sg, so, m = p["dyc"], p["sg"], p["so"], p["m"]
    for n in STATE:
        yield n, sg[n].cpu().numpy(), so[n]
    for n in AUX:
        yield n, dyc.rk2_field(n).cpu().numpy(), m[n]


def _drift(p):
    out = {}
    for n, a, b in _fields(p):
        sl = (Ellipsis,) + tuple(H.interior(p["d"], STAG[n]))
        a = a[sl]; b = b[sl]
        out[n] = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
    return out


def _budgets(p):
    """Order-independent properties of the device state: sum_k uh = uhbt to ETA_TOLERANCE, volume unchanged to round-off."""
    d, M, dt, sg, dyc = p["d"], p["M"], p["dt"], p["sg"], p["dyc"]
    sl = H.interior(d, "h")
    A = M[G["areaT"]][sl]
    vol0 = (p["inp"]["h"][(Ellipsis,) + sl] * A).sum(); vol1 = (sg["h"].cpu().numpy()[(Ellipsis,) + sl] * A).sum()
    assert abs(vol1 / vol0 - 1.0) < 1e-13, vol1 / vol0 - 1.0
    uh = sg["uh"].cpu().numpy(); uhbt = dyc.rk2_field("uhbt").cpu().numpy()
    IA = M[G["IareaT"]]
    su = H.interior(d, "u")
    Imin = np.minimum(IA, np.roll(IA, -1, axis=1))[su]      # the larger of the two eta changes a transport error makes (:1171-1178)
    err = np.abs(uh.sum(0) - uhbt)[su] * dt * Imin * (M[G["mask2dCu"]][su] > 0)
    assert err.max() <= p["cont"].tol_eta * 1.000001, (err.max(), p["cont"].tol_eta)
    return float(err.max())


def _report(name, rows, extra=None):
    worst = {n: max(r[n] for r in rows) for n in rows[0]}
    line = f"[sum-order drift] {name}: after {len(rows)} steps max over fields {max(rows[-1].values()):.2e} of range; per step " + \
           " ".join(f"{max(r.values()):.1e}" for r in rows)
    print(line)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):
        path = os.path.join(out, "sum_order_drift.json")
        try:
            doc = json.load(open(path))
        except Exception:
            doc = {}
        doc[name] = dict(per_step_max_over_fields=[max(r.values()) for r in rows], last_step_per_field=rows[-1], worst_per_field=worst,
                         bound=BOUND, **(extra or {}))
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("case,nsteps", [("benchmark_small_75", 10), ("island_basin_75", 10), ("benchmark_small_75_pow", 10),
                                         ("island_basin_75_pow", 10)])
def test_default_device_order_stays_within_the_stated_bound_of_the_reference_order(orc, case, nsteps):
    """..._pow: the same ten steps on btstep's DEFAULT drag path (BT_STRONG_DRAG = False: bt_rem = av_rem**(1/nstep), the one
    expression of the path where the device's pow and libm's may differ in the last bit), so that the stated bound covers the
    configuration the benchmark runs: sum order + pow together, against the REFERENCE-order oracle with libm's pow."""
    pow_path = case.endswith("_pow")
    cfg = dict(benchmark_small_75=lambda: H.benchmark_small(nk=75), island_basin_75=lambda: H.island_basin(nk=75))[case.replace("_pow", "")]()
    p = _pair(orc, cfg, abi.SUM_TREE16, abi.SUM_REFERENCE, strong_drag=0 if pow_path else 1)
    rows = []
    for n in range(nsteps):
        p["step"](n)
        rows.append(_drift(p))
    eta_err = _budgets(p)
    _report(case, rows, dict(sum_k_uh_minus_uhbt_as_eta_change=eta_err, tol_eta=p["cont"].tol_eta))
    bad = {n: v for n, v in rows[-1].items() if v > BOUND}
    assert not bad, bad
    p["dyc"].close()


def test_default_device_order_on_config2_grid_one_step(orc):
    """BASELINE.json configs[2]'s 360 x 180 x 75 grid, one step: the same bound."""
    p = _pair(orc, H.benchmark_360(), abi.SUM_TREE16, abi.SUM_REFERENCE)
    p["step"](0)
    rows = [_drift(p)]
    eta_err = _budgets(p)
    _report("benchmark_360x180x75", rows, dict(sum_k_uh_minus_uhbt_as_eta_change=eta_err, tol_eta=p["cont"].tol_eta))
    bad = {n: v for n, v in rows[-1].items() if v > BOUND}
    assert not bad, bad
    p["dyc"].close()


def test_reference_order_on_the_device_is_bit_identical_over_ten_steps(orc):
    """MOM6X_SUMS=exact (sum_order = MOM6X_SUM_REFERENCE): ten steps at nk = 75 with every callee on the device equal the
    REFERENCE-order oracle in every bit, zeros of opposite sign included."""
    p = _pair(orc, H.benchmark_small(nk=75), abi.SUM_REFERENCE, abi.SUM_REFERENCE)
    for n in range(10):
        p["step"](n)
    for n, a, b in _fields(p):
        H.assert_bitwise(a, b, n, H.interior(p["d"], STAG[n]), signed_zero_ok=False)
    _budgets(p)
    p["dyc"].close()


def test_fused_multiply_add_arithmetic_over_ten_steps(orc):
    """sum_order = MOM6X_SUM_TREE16_FMA (opt-in: the tree's sums and fused multiply-adds at fixed sites of the flux and the PPM
    edge formulas, include/mom6x.h): ten steps at nk = 75 with every callee on the device (a) equal the oracle's restatement of the
    same sites in every bit, and (b) stay within the stated bound of the REFERENCE-order, un-fused oracle -- the reference's own
    arithmetic.  The observed drift goes to gpurun_out/sum_order_drift.json under "fma_benchmark_small_75"."""
    cfg = H.benchmark_small(nk=75)
    p = _pair(orc, cfg, abi.SUM_TREE16_FMA, abi.SUM_TREE16_FMA)
    for n in range(10):
        p["step"](n)
    for n, a, b in _fields(p):
        H.assert_bitwise(a, b, n, H.interior(p["d"], STAG[n]), signed_zero_ok=False)
    _budgets(p)
    p["dyc"].close()
    p = _pair(orc, cfg, abi.SUM_TREE16_FMA, abi.SUM_REFERENCE, strong_drag=0)
    rows = []
    for n in range(10):
        p["step"](n)
        rows.append(_drift(p))
    eta_err = _budgets(p)
    _report("fma_benchmark_small_75_pow", rows, dict(sum_k_uh_minus_uhbt_as_eta_change=eta_err, tol_eta=p["cont"].tol_eta))
    bad = {n: v for n, v in rows[-1].items() if v > BOUND}
    assert not bad, bad
    p["dyc"].close()
