"""The reference's own differential checks run ON THE DEVICE, at full size (.testing/Makefile:399-408; rules :608-615):

  * test.rotate (ROTATE_INDEX, INDEX_TURNS = 1): a closed basin and its quarter turn (cell (i, j) -> (nj-1-j, i), u' = -v, v' = u,
    FIRST_DIRECTION flipped) through two baroclinic steps with every callee on the device and one advect_tracer: every
    prognostic field of the turned run equals the turned field of the original run BIT FOR BIT.  The zonal and the
    meridional kernels of the device are different programs (DIR = 0 / 1 instantiations with different staging, march and
    work-group shapes): the turn holds each of them to the other.
  * test.dim.t / .l / .h / .z / .r: units of time, length, thickness, depth or density scaled by 2**11 (inputs, metrics,
    mom6x_vgrid, every dimensional parameter): the unscaled answers equal the unscaled run's bit for bit.
  * test.nan: the device's work arrays start as NaNs (MOM6X_POISON_WORK, mom6_amd/csrc/ctx.hip): same bits as with zeros.

None of them passes through oracle/: this is the parity evidence that does not depend on the builder's restatement of the
reference, and the only strong check the paths taken at 1440 x 1080 x 75 (large-tile barotropic sub-cycle, march lengths of
the mass-flux kernels, XCD orders, two-part launches) can get -- the oracle needs minutes per step there.  The same checks run
on the oracle in tests/test_oracle_invariants_cpu.py (whose Turn is reused here); both orders of the column sums."""
import numpy as np
import pytest

from mom6_amd import abi, grid
from tests import helpers as H
from tests.test_oracle_invariants_cpu import Turn

pytestmark = pytest.mark.gpu
G = abi.G
SIZES = {"small": (60, 44, 12), "full": (1440, 1080, 75)}
STAG = dict(u="u", v="v", h="h", uh="u", vh="v", uhtr="u", vhtr="v", eta_av="h", tr0="h", tr1="h")


class TTurn(Turn):
    """Turn for torch tensors on the device (3-D arrays of 1 GB are turned where they live)."""

    def _put(self, a, src_j, src_i, dst_j, dst_i, sign=1.0):
        if isinstance(a, np.ndarray):
            return super()._put(a, src_j, src_i, dst_j, dst_i, sign)
        import torch
        d, dr = self.d, self.dr
        dev = a.device
        out = torch.zeros(tuple(a.shape[:-2]) + tuple(dr.shape2()), dtype=a.dtype, device=dev)
        sj = torch.as_tensor(src_j + d.joff, device=dev)[:, None]; si = torch.as_tensor(src_i + d.ioff, device=dev)[None, :]
        dj = torch.as_tensor(dst_j + dr.joff, device=dev)[:, None]; di = torch.as_tensor(dst_i + dr.ioff, device=dev)[None, :]
        g = a[..., sj, si].transpose(-1, -2)
        out[..., dj, di] = -g if sign < 0 else g
        return out.contiguous()


def _basin(size):
    ni, nj, nk = SIZES[size]
    gg = grid.GlobalGrid(ni, nj, kind="spherical", lon0=0.0, lat0=-65.0, dlon=360.0 / ni, dlat=130.0 / nj, reentrant_x=False,
                         depth_fn=grid.bowl_depth(ni, nj, 4000.0, rim=2))
    return gg.tile(nk)


def _bits_equal(a, b, name, sl, zeros_of_either_sign=False):
    """Device tensors, bit for bit on the slices sl."""
    import torch
    a = a[(Ellipsis,) + tuple(sl)]; b = b[(Ellipsis,) + tuple(sl)]
    assert a.shape == b.shape, (name, a.shape, b.shape)
    ne = a.contiguous().view(torch.int64) != b.contiguous().view(torch.int64)
    if zeros_of_either_sign:   # a quarter turn negates a velocity component: -(+0) = -0 is the turn itself
        ne = ne & ~((a == 0.0) & (b == 0.0))
    n = int(ne.sum().item())
    if n:
        diff = torch.where(ne, (a - b).abs(), torch.zeros_like(a))
        raise AssertionError(f"{name}: {n} words differ; max|diff| = {diff.max().item():.3e} of {b.abs().max().item():.3e}")
    assert bool(torch.isfinite(a).all()), name


def _inputs(d, Md, dev):
    """The benchmark's recipe (bench.build_model) on a closed basin, with a meridional wind as well."""
    from mom6_amd import synth_dev
    h, u, v = synth_dev.make_state(d, Md, u_max=0.5, h_pert=0.01)
    return dict(
        h=h, u=u, v=v,
        taux=(0.1 * synth_dev.smooth_field(d, dev, 41, ox=1.0, oy=0.5) * Md[G["mask2dCu"]]).contiguous(),
        tauy=(0.05 * synth_dev.smooth_field(d, dev, 42, ox=0.5, oy=1.0) * Md[G["mask2dCv"]]).contiguous(),
        kbu=(2.0e-3 * (1.0 + 0.5 * synth_dev.smooth_field(d, dev, 91, ox=1.0, oy=0.5)) * Md[G["mask2dCu"]]).contiguous(),
        kbv=(2.0e-3 * (1.0 + 0.5 * synth_dev.smooth_field(d, dev, 92, ox=0.5, oy=1.0)) * Md[G["mask2dCv"]]).contiguous(),
        tr0=(10.0 + 5.0 * synth_dev.smooth_field(d, dev, 71, nk=d.nk, ox=0.5, oy=0.5)).contiguous(),
        tr1=(35.0 + 0.5 * synth_dev.smooth_field(d, dev, 82, nk=d.nk)).contiguous())


def _run(d, M, first_direction, inp, nsteps=2, dt=900.0, full_callees=True, GV=None, scaled=None, coefs=None):
    """A device model of the tile, `nsteps` baroclinic steps, then advect_tracer of two PPM tracers with the accumulated
    transports.  full_callees: vertvisc_coef and horizontal_viscosity inside the step (the benchmark's workload); otherwise the
    coupling coefficients `coefs` = (a_u, a_v, h_u, h_v) come from outside.  Returns clones of the results (the model is closed)."""
    import torch
    import bench
    from mom6_amd.dycore import Dycore
    GV = GV or abi.vgrid_default()
    s = scaled or dict(T=1.0, L=1.0, H=1.0, Z=1.0, R=1.0)
    dyc = Dycore(d, M, GV, first_direction)
    cont = abi.continuity_params_default(d.nk, GV.Angstrom_H)
    cont.tol_vel = cont.tol_vel * s["L"] / s["T"]
    dyc.continuity_init(cont)
    dyc.barotropic_init(abi.barotropic_params_default(20.0 * s["T"]))
    dyc.CoriolisAdv_init(abi.coriolis_params_default())
    Rlay, gp = abi.layer_densities(d.nk)
    Rlay = np.ascontiguousarray(Rlay * s["R"]); gp = np.ascontiguousarray(gp * s["L"] * s["L"] / (s["Z"] * s["T"] * s["T"]))
    dyc.PressureForce_init(abi.pgf_params_default(GV.Rho0), Rlay, gp)
    dyc.initialize_dyn_split_RK2(abi.rk2_params_default())
    st = dict(u=inp["u"].clone(), v=inp["v"].clone(), h=inp["h"].clone(), uh=dyc.zeros3(), vh=dyc.zeros3(), uhtr=dyc.zeros3(),
              vhtr=dyc.zeros3(), eta_av=dyc.zeros2())
    keep = []
    if full_callees:
        dyc.vertvisc_init(abi.vertvisc_params_default(Kv=1.0e-4, Hmix=20.0, Hbbl=10.0))
        bbu = torch.full_like(inp["kbu"], 10.0); bbv = torch.full_like(inp["kbv"], 10.0)
        keep += [bbu, bbv]
        dyc.vertvisc_set_visc(inp["kbu"], inp["kbv"], bbu, bbv)
        dyc.vertvisc_coef(st["u"], st["v"], st["h"], dt)
        dyc.hor_visc_init(bench.hor_visc_params(abi, dt))
    else:
        dyc.vertvisc_set_coef(*coefs)
    torch.cuda.synchronize()
    dyc.dyn_split_RK2_new_run(st["u"], st["v"], st["h"], st["uh"], st["vh"], dt)
    for n in range(nsteps):
        dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"], inp["taux"],
                                   inp["tauy"], dt, calc_dtbt=(n == 0))
    dyc.sync()
    out = {n: st[n].clone() for n in st}
    if "tr0" in inp:
        trs = [inp["tr0"].clone(), inp["tr1"].clone()]
        torch.cuda.synchronize()   # (the clones are made on torch's stream, the model has its own)
        dyc.tracer_advect_init(dt, 2)
        its = dyc.advect_tracer(st["h"], st["uhtr"], st["vhtr"], nsteps * dt, trs, [2, 2])
        dyc.sync()
        assert 1 <= its <= 8
        out["tr0"], out["tr1"] = trs
    torch.cuda.synchronize()
    dyc.close()
    return out


@pytest.mark.parametrize("size", ["small", "full"])
def test_rotate_two_steps_and_advect_tracer_on_the_device(size, sums):
    import torch
    dev = torch.device("cuda", 0)
    d, M = _basin(size)
    T = TTurn(d); dr = T.dr; Mr = T.metrics(M)
    Md = torch.as_tensor(M, device=dev)
    inp = _inputs(d, Md, dev)
    a = _run(d, M, 0, inp)
    assert a["u"].abs().max().item() > 1e-3 and bool(torch.isfinite(a["h"]).all())
    # the turned problem: u' = -v on what were v faces, v' = u; face scalars only change their staggering
    inr = dict(u=T.v_to_u(inp["v"]), v=T.u_to_v(inp["u"]), h=T.h(inp["h"]), taux=T.v_to_u(inp["tauy"]), tauy=T.u_to_v(inp["taux"]),
               kbu=T.v_to_u(inp["kbv"], 1.0), kbv=T.u_to_v(inp["kbu"], 1.0), tr0=T.h(inp["tr0"]), tr1=T.h(inp["tr1"]))
    del inp
    b = _run(dr, Mr, 1, inr)
    del inr
    su, sv, sh = H.interior(dr, "u"), H.interior(dr, "v"), H.interior(dr, "h")
    for x, y in (("u", "v"), ("uh", "vh"), ("uhtr", "vhtr")):
        _bits_equal(b[x], T.v_to_u(a[y]), f"rotate:{x}' = -{y}", su, True)
        _bits_equal(b[y], T.u_to_v(a[x]), f"rotate:{y}' = {x}", sv, True)
    for x in ("h", "eta_av", "tr0", "tr1"):
        _bits_equal(b[x], T.h(a[x]), "rotate:" + x, sh, True)


def _scaled_problem(d, M, inp, dim, p=11):
    """Everything of the problem in units scaled by 2**p in one dimension (MOM_unit_scaling.F90:92, MOM_verticalGrid.F90:157)."""
    s = 2.0 ** p
    sc = dict(T=1.0, L=1.0, H=1.0, Z=1.0, R=1.0)
    if dim:
        sc[dim.upper()] = s
    T_, L, Hs, Z, R = sc["T"], sc["L"], sc["H"], sc["Z"], sc["R"]
    M2 = M.copy()
    for n in abi.METRICS:
        if n.startswith(("dx", "dy")): M2[G[n]] = M[G[n]] * L
        elif n.startswith(("Idx", "Idy")): M2[G[n]] = M[G[n]] / L
        elif n.startswith("area"): M2[G[n]] = M[G[n]] * L * L
        elif n.startswith("Iarea"): M2[G[n]] = M[G[n]] / (L * L)
        elif n == "bathyT": M2[G[n]] = M[G[n]] * Z
        elif n == "CoriolisBu": M2[G[n]] = M[G[n]] / T_
        elif n == "Coriolis2Bu": M2[G[n]] = M[G[n]] / (T_ * T_)
    GV = abi.vgrid_default(); GV2 = abi.vgrid_default()
    GV2.g_Earth = GV.g_Earth * L * L / (Z * T_ * T_)
    GV2.Rho0 = GV.Rho0 * R
    GV2.Angstrom_H = GV.Angstrom_H * Hs; GV2.H_subroundoff = GV.H_subroundoff * Hs; GV2.dZ_subroundoff = GV.dZ_subroundoff * Z
    GV2.H_to_Z = GV.H_to_Z * Z / Hs; GV2.Z_to_H = GV.Z_to_H * Hs / Z
    GV2.H_to_RZ = GV.H_to_RZ * R * Z / Hs; GV2.RZ_to_H = GV.RZ_to_H * Hs / (R * Z)
    in2 = dict(u=(inp["u"] * (L / T_)).contiguous(), v=(inp["v"] * (L / T_)).contiguous(), h=(inp["h"] * Hs).contiguous(),
               taux=(inp["taux"] * (R * Z * L / (T_ * T_))).contiguous(), tauy=(inp["tauy"] * (R * Z * L / (T_ * T_))).contiguous())
    coefs = tuple((c * (Hs / T_) if q < 2 else c * Hs).contiguous() for q, c in enumerate(inp["coefs"]))
    unscale = dict(u=T_ / L, v=T_ / L, h=1.0 / Hs, uh=T_ / (Hs * L * L), vh=T_ / (Hs * L * L), uhtr=1.0 / (Hs * L * L), vhtr=1.0 / (Hs * L * L),
                   eta_av=1.0 / Hs)
    return np.ascontiguousarray(M2), GV2, sc, in2, coefs, unscale


@pytest.mark.parametrize("size", ["small", "full"])
def test_dim_rescaling_on_the_device(size):
    """Two baroclinic steps (continuity x3, CorAdCalc, PressureForce, btstep and its sub-cycle, vertvisc, the RK2 glue; the
    coupling coefficients from outside, as in the oracle's version of this test) in unscaled units and with the unit of time,
    length, thickness, depth or density scaled by 2**11 (one model per dimension, all against the one unscaled run)."""
    import torch
    dev = torch.device("cuda", 0)
    d, M = _basin(size)
    Md = torch.as_tensor(M, device=dev)
    inp = _inputs(d, Md, dev)
    for n in ("tr0", "tr1", "kbu", "kbv"):
        inp.pop(n)
    a = torch.zeros((d.nk + 1,) + tuple(Md.shape[1:]), dtype=torch.float64, device=dev); a[1:] = 1e-5; a[d.nk] = 3e-4
    hu = torch.clamp(inp["h"], min=1e-9)
    inp["coefs"] = ((a * Md[G["mask2dCu"]][None]).contiguous(), (a * Md[G["mask2dCv"]][None]).contiguous(), hu.contiguous(), hu.clone())
    del a
    ref = None
    for dim in ("", "t", "l", "h", "z", "r"):
        M2, GV2, sc, in2, coefs, unscale = _scaled_problem(d, M, inp, dim)
        out = _run(d, M2, 0, in2, dt=900.0 * sc["T"], full_callees=False, GV=GV2, scaled=sc, coefs=coefs)
        res = {n: out[n] * unscale[n] for n in unscale}
        del out, in2, coefs
        if ref is None:
            ref = res
            assert ref["u"].abs().max().item() > 1e-3
            continue
        for n in ref:
            _bits_equal(res[n], ref[n], f"dim.{dim}:{n}", H.interior(d, STAG[n]))


@pytest.mark.parametrize("size", ["small", "full"])
def test_nan_poisoned_work_arrays_on_the_device(size, sums, monkeypatch):
    """Every work array of the device starts as NaNs: the two steps and the tracer advection give the same bits."""
    import torch
    dev = torch.device("cuda", 0)
    d, M = _basin(size)
    Md = torch.as_tensor(M, device=dev)
    inp = _inputs(d, Md, dev)
    monkeypatch.setenv("MOM6X_POISON_WORK", "0")
    a = _run(d, M, 0, inp)
    monkeypatch.setenv("MOM6X_POISON_WORK", "1")
    b = _run(d, M, 0, inp)
    for n in a:
        _bits_equal(b[n], a[n], "nan:" + n, H.interior(d, STAG[n]))
