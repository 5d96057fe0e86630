"""The reference's own differential invariants applied to the oracle -- the only pin the hot path has (the reference
holds no numbers for it and cannot be built here; SURVEY.md section 4, oracle/orc_common.h):

  * rotate   (.testing/Makefile:602-616, ROTATE_INDEX): a quarter turn of the index space.  u' = v, v' = -u on the
             turned grid with FIRST_DIRECTION flipped must give the turned answers BIT FOR BIT -- the reference's
             parenthesisation ((a+b)+(c+d), symmetric stencils) exists for exactly this, and a restatement that
             mixes up an index, a staggering or the order of a sum fails it;
  * dim.t / dim.l / dim.h / dim.z / dim.r (.testing/Makefile:151, README.rst): rescaling time, length, thickness, depth
             or density units by 2**11 changes no bit of the answers (after unscaling);
  * conservation / known answers of the tracer advection and of the barotropic solver.

All of them run through the WHOLE baroclinic step (continuity, CorAdCalc, PressureForce, btstep with its sub-cycle,
vertvisc) where the routine allows it."""
import numpy as np
import pytest

from mom6_amd import abi, synth
from tests import helpers as H

G = abi.G
H_M = ("mask2dT", "dxT", "dyT", "IdxT", "IdyT", "areaT", "IareaT", "bathyT")
U_M = ("mask2dCu", "dxCu", "dyCu", "IdxCu", "IdyCu", "areaCu", "IareaCu", "dy_Cu")
V_M = ("mask2dCv", "dxCv", "dyCv", "IdxCv", "IdyCv", "areaCv", "IareaCv", "dx_Cv")
Q_M = ("mask2dBu", "dxBu", "dyBu", "IdxBu", "IdyBu", "areaBu", "IareaBu", "CoriolisBu", "Coriolis2Bu")
SWAP = {"dxT": "dyT", "dyT": "dxT", "IdxT": "IdyT", "IdyT": "IdxT", "dxBu": "dyBu", "dyBu": "dxBu", "IdxBu": "IdyBu", "IdyBu": "IdxBu",
        # a u-point becomes a v-point and the x and y lengths trade places
        "dxCu": "dyCv", "dyCu": "dxCv", "IdxCu": "IdyCv", "IdyCu": "IdxCv", "dy_Cu": "dx_Cv", "mask2dCu": "mask2dCv", "areaCu": "areaCv", "IareaCu": "IareaCv",
        "dxCv": "dyCu", "dyCv": "dxCu", "IdxCv": "IdyCu", "IdyCv": "IdxCu", "dx_Cv": "dy_Cu", "mask2dCv": "mask2dCu", "areaCv": "areaCu", "IareaCv": "IareaCu"}


class Turn:
    """A quarter turn (+90 degrees) of a closed-basin tile: cell (i, j) -> (nj-1-j, i); the u face (I, j) becomes the
    v face (nj-1-j, I) with v' = u; the v face (i, J) becomes the u face (nj-2-J, i) with u' = -v; the vertex (I, J)
    becomes (nj-2-J, I).  Works on whole pitched arrays, halos included."""

    def __init__(self, d):
        self.d = d
        self.dr = abi.dims_init(d.nj, d.ni, d.nk, d.halo)
        w = d.halo
        self.ih = np.arange(-w, d.ni + w); self.jh = np.arange(-w, d.nj + w)            # h-point data domain
        self.iB = np.arange(-w - 1, d.ni + w); self.jB = np.arange(-w - 1, d.nj + w)    # face / vertex ranges

    def _put(self, a, src_j, src_i, dst_j, dst_i, sign=1.0):
        d, dr = self.d, self.dr
        out = np.zeros(a.shape[:-2] + dr.shape2())
        out[..., (dst_j + dr.joff)[:, None], (dst_i + dr.ioff)[None, :]] = sign * np.swapaxes(
            a[..., (src_j + d.joff)[:, None], (src_i + d.ioff)[None, :]], -1, -2)
        return np.ascontiguousarray(out)

    # source ranges (j, i) -> destination (j' = i-like, i' = nj-1-j-like); the swapaxes puts the old i on the new j axis
    def h(self, a):
        return self._put(a, self.jh, self.ih, self.ih, self.d.nj - 1 - self.jh)

    def u_to_v(self, a, sign=1.0):     # u(I, j) -> v'(i' = nj-1-j, J' = I)
        return self._put(a, self.jh, self.iB, self.iB, self.d.nj - 1 - self.jh, sign)

    def v_to_u(self, a, sign=-1.0):    # v(i, J) -> u'(I' = nj-2-J, j' = i)
        return self._put(a, self.jB, self.ih, self.ih, self.d.nj - 2 - self.jB, sign)

    def q(self, a):                    # q(I, J) -> q'(I' = nj-2-J, J' = I)
        return self._put(a, self.jB, self.iB, self.iB, self.d.nj - 2 - self.jB)

    def metrics(self, M):
        Mr = np.zeros((abi.G_COUNT,) + self.dr.shape2())
        for n in H_M:
            Mr[G[SWAP.get(n, n)]] = self.h(M[G[n]])
        for n in U_M:
            Mr[G[SWAP[n]]] = self.u_to_v(M[G[n]])
        for n in V_M:
            Mr[G[SWAP[n]]] = self.v_to_u(M[G[n]], sign=1.0)
        for n in Q_M:
            Mr[G[SWAP.get(n, n)]] = self.q(M[G[n]])
        return np.ascontiguousarray(Mr)

    # back: a turned h / u' / v' array as it would look on the original grid
    def h_back(self, a):
        return Turn(self.dr)._h3(a)

    def _h3(self, a):      # three more quarter turns = the inverse
        t = self
        for _ in range(3):
            a = t.h(a); t = Turn(t.dr)
        return a


def _turn_back(T, ur, vr, hr):
    """(u', v', h') on the turned grid -> (u, v, h) on the original one: u = v', v = -u'."""
    d, dr = T.d, T.dr
    Tb = Turn(dr)
    # three further quarter turns bring a field home; per turn (u, v) -> (-v, u)
    u, v, h = ur, vr, hr
    t = Tb
    for _ in range(3):
        u, v, h = t.v_to_u(v), t.u_to_v(u), t.h(h)
        t = Turn(t.dr)
    return u, v, h


def _state(cfg, **kw):
    gg, d, M = cfg
    h, u, v = synth.make_state(d, M, **kw)
    return d, M, h, u, v


def test_turn_is_its_own_fourth_root():
    d, M, h, u, v = _state(H.benchmark_small())
    T = Turn(d)
    u2, v2, h2 = _turn_back(T, T.v_to_u(v), T.u_to_v(u), T.h(h))
    sh = H.interior(d, "h", d.halo)
    np.testing.assert_array_equal(h2[(Ellipsis,) + sh], h[(Ellipsis,) + sh])
    su, sv = H.interior(d, "u", 3), H.interior(d, "v", 3)
    np.testing.assert_array_equal(u2[(Ellipsis,) + su], u[(Ellipsis,) + su])
    np.testing.assert_array_equal(v2[(Ellipsis,) + sv], v[(Ellipsis,) + sv])
    Mr = T.metrics(M)
    assert np.array_equal(Mr[G["dxT"]], T.h(M[G["dyT"]])) and np.array_equal(Mr[G["dyCu"]], T.v_to_u(M[G["dxCv"]], 1.0))
    assert np.abs(Mr[G["mask2dCu"]]).sum() == np.abs(M[G["mask2dCv"]]).sum() > 0


@pytest.mark.parametrize("sum_order,flags", [(abi.SUM_REFERENCE, {}), (abi.SUM_TREE16, {}), (abi.SUM_REFERENCE, dict(vol_CFL=1)),
                                             (abi.SUM_TREE16, dict(aggress_adjust=1, vol_CFL=1)), (abi.SUM_REFERENCE, dict(aggress_adjust=1))])
@pytest.mark.parametrize("first_direction", [0, 1])
def test_rotate_continuity_and_CorAdCalc(orc, first_direction, sum_order, flags):
    """continuity_PPM with the Newton adjustment towards uhbt / vhbt and the BT_cont fits, then CorAdCalc, on a grid and its
    quarter turn: every output of the turned run, turned back, equals the original run bit for bit."""
    d, M, h, u, v = _state(H.benchmark_small(), thin_frac=0.1)
    if flags:
        M = H.narrowed_faces(d, M)   # (open face widths that differ from the cell widths: what the volume-based CFL number sees)
    GV = abi.vgrid_default(); dt = 900.0
    T = Turn(d); dr = T.dr; Mr = T.metrics(M)
    vr_u = np.ascontiguousarray(np.clip(0.5 + 0.6 * synth.smooth_field(d, 11, nk=d.nk, ox=1.0, oy=0.5), 0.0, 1.0))
    vr_v = np.ascontiguousarray(np.clip(0.5 + 0.6 * synth.smooth_field(d, 12, nk=d.nk, ox=0.5, oy=1.0), 0.0, 1.0))
    uhbt = np.ascontiguousarray(30.0 * synth.smooth_field(d, 13, ox=1.0, oy=0.5) * M[G["mask2dCu"]] * M[G["dy_Cu"]] * 1e-3)
    vhbt = np.ascontiguousarray(30.0 * synth.smooth_field(d, 14, ox=0.5, oy=1.0) * M[G["mask2dCv"]] * M[G["dx_Cv"]] * 1e-3)

    def run(d_, M_, fd, u_, v_, h_, ub, vb, vru, vrv):
        CS = abi.continuity_params_default(d_.nk, GV.Angstrom_H); CS.sum_order = sum_order
        for k_, v__ in flags.items():   # CONT_PPM_AGGRESS_ADJUST / CONT_PPM_VOLUME_BASED_CFL: the zonal and meridional
            setattr(CS, k_, v__)        # branches (:651-716 / :1544-1608) are each other's quarter turn
        z = lambda: np.zeros_like(h_)
        o = dict(h=z(), uh=z(), vh=z(), u_cor=z(), v_cor=z())
        bt = orc.new_bt_cont(d_)
        orc.continuity_PPM(d_, M_, GV, CS, fd, u_, v_, h_, o["h"], o["uh"], o["vh"], dt, uhbt=ub, vhbt=vb, visc_rem_u=vru, visc_rem_v=vrv,
                           u_cor=o["u_cor"], v_cor=o["v_cor"], BT_cont=bt)
        o["CAu"], o["CAv"] = z(), z()
        orc.CorAdCalc(d_, M_, GV, abi.coriolis_params_default(), u_, v_, o["h"], o["uh"], o["vh"], o["CAu"], o["CAv"])
        return o, bt

    o, bt = run(d, M, first_direction, u, v, h, uhbt, vhbt, vr_u, vr_v)
    # the turned problem: u' = -v (on what were v faces), v' = u; transports likewise; visc_rem is a scalar on faces
    orr, btr = run(dr, Mr, 1 - first_direction, T.v_to_u(v), T.u_to_v(u), T.h(h), T.v_to_u(vhbt), T.u_to_v(uhbt),
                   T.v_to_u(vr_v, 1.0), T.u_to_v(vr_u, 1.0))
    su, sv, sh = H.interior(d, "u"), H.interior(d, "v"), H.interior(d, "h")
    for a, b in (("uh", "vh"), ("u_cor", "v_cor"), ("CAu", "CAv")):
        ub, vb, hb = _turn_back(T, orr[a], orr[b], orr["h"])
        H.assert_bitwise(ub, o[a], "rotate:" + a, su); H.assert_bitwise(vb, o[b], "rotate:" + b, sv)
        H.assert_bitwise(hb, o["h"], "rotate:h", sh)
    # BT_cont: the east side of a turned u face is the north side of the original v face, and so on
    # (compared on the turned grid: a face area is a scalar that only changes its staggering; east of a u face is north
    # of the v face it becomes, north of a v face is WEST of the u face it becomes -- u' = -v)
    for n_orig, n_turn in (("FA_u_EE", "FA_v_NN"), ("FA_u_E0", "FA_v_N0"), ("FA_u_W0", "FA_v_S0"), ("FA_u_WW", "FA_v_SS"),
                           ("FA_v_NN", "FA_u_WW"), ("FA_v_N0", "FA_u_W0"), ("FA_v_S0", "FA_u_E0"), ("FA_v_SS", "FA_u_EE")):
        want = T.u_to_v(bt[n_orig], 1.0) if n_orig.startswith("FA_u") else T.v_to_u(bt[n_orig], 1.0)
        sl = H.interior(dr, "v" if n_orig.startswith("FA_u") else "u")
        H.assert_bitwise(btr[n_turn], want, "rotate:BT_cont%" + n_orig, sl)
    H.assert_bitwise(btr["h_v"], T.u_to_v(bt["h_u"], 1.0), "rotate:BT_cont%h_u", H.interior(dr, "v"))
    H.assert_bitwise(btr["h_u"], T.v_to_u(bt["h_v"], 1.0), "rotate:BT_cont%h_v", H.interior(dr, "u"))


SCHEMES = {
    "AH90": dict(Coriolis_Scheme=abi.ARAKAWA_HSU90),
    "AL81": dict(Coriolis_Scheme=abi.ARAKAWA_LAMB81),
    "AL81_bound": dict(Coriolis_Scheme=abi.ARAKAWA_LAMB81, bound_Coriolis=1),
    "AL_BLEND": dict(Coriolis_Scheme=abi.AL_BLEND),
    "AL_BLEND_3_0.5": dict(Coriolis_Scheme=abi.AL_BLEND, F_eff_max_blend=3.0, wt_lin_blend=0.5),
    "ROBUST": dict(Coriolis_Scheme=abi.ROBUST_ENSTRO),
    "ROBUST_upwind": dict(Coriolis_Scheme=abi.ROBUST_ENSTRO, PV_Adv_Scheme=abi.PV_ADV_UPWIND1),
}


def _coriolis_case(thin_frac=0.1):
    d, M, h, u, v = _state(H.benchmark_small(), thin_frac=thin_frac)
    h = H.roughen(h)
    uh = np.ascontiguousarray(u * 1.0e5 * (1 + 0.1 * synth.smooth_field(d, 7, nk=d.nk)) * M[G["mask2dCu"]])
    vh = np.ascontiguousarray(v * 1.0e5 * (1 + 0.1 * synth.smooth_field(d, 8, nk=d.nk)) * M[G["mask2dCv"]])
    return d, M, h, u, v, uh, vh


def _blend_weight_census(d, M, h, F_eff_max=4.0, wt_lin=0.125):
    """Which of the seven regimes of ARAKAWA_LAMB_BLEND's two weights (:559-573) the cells of a state fall in."""
    A = (M[G["mask2dT"]] * M[G["areaT"]])[None]
    Ah = A * h
    r = lambda a, di, dj: np.roll(np.roll(a, -di, axis=-1), -dj, axis=-2)
    hAu = 0.5 * (Ah + r(Ah, 1, 0)); hAv = 0.5 * (Ah + r(Ah, 0, 1))
    Aq = (A + r(A, 1, 1)) + (r(A, 1, 0) + r(A, 0, 1))
    Ihq = Aq / ((hAu + r(hAu, 0, 1)) + (hAv + r(hAv, 1, 0)) + 1e-40)
    c = [Ihq, r(Ihq, -1, 0), r(Ihq, 0, -1), r(Ihq, -1, -1)]
    sl = (Ellipsis,) + tuple(H.interior(d, "h"))
    mn = np.minimum.reduce(c)[sl]; mx = np.maximum.reduce(c)[sl]
    ok = (mn > 0) & (M[G["mask2dT"]][tuple(H.interior(d, "h"))] > 0)[None]
    rat = (mx[ok] / mn[ok]) - 1.0
    Fe = F_eff_max - 2.0; rl = 1.5 * Fe / wt_lin
    out = set()
    for n, m in enumerate([rat <= Fe, (rat > Fe) & (rat < 1.5 * Fe), rat >= 1.5 * Fe, rat <= 1.5 * Fe, (rat > 1.5 * Fe) & (rat <= rl),
                           (rat > rl) & (rat < 2 * rl), rat >= 2 * rl]):
        if m.any():
            out.add(n)
    return out


def _corad(orc, d, M, mods, u, v, h, uh, vh):
    CS = abi.coriolis_params_default()
    for k, val in mods.items():
        setattr(CS, k, val)
    CAu, CAv = np.zeros_like(h), np.zeros_like(h)
    orc.CorAdCalc(d, M, abi.vgrid_default(), CS, u, v, h, uh, vh, CAu, CAv)
    return CAu, CAv


@pytest.mark.parametrize("name", sorted(SCHEMES))
def test_rotate_CorAdCalc_every_scheme(orc, name):
    """The Coriolis schemes beyond the default (ARAKAWA_HSU90, ARAKAWA_LAMB81, ARAKAWA_LAMB_BLEND, ROBUST_ENSTRO with both
    PV_ADV_SCHEMEs; MOM_CoriolisAdv.F90:523-588, :683-721, :796-845) on a grid and its quarter turn: the u and v halves of
    each scheme are each other's image (the Arakawa weights a, b, c, d and ep_u, ep_v permute), so CAu / CAv of the turned
    run, turned back, equal the original to round-off -- the pairs of products are not always added in the image's order."""
    d, M, h, u, v, uh, vh = _coriolis_case()
    T = Turn(d); dr = T.dr; Mr = T.metrics(M)
    CAu, CAv = _corad(orc, d, M, SCHEMES[name], u, v, h, uh, vh)
    ru, rv = _corad(orc, dr, Mr, SCHEMES[name], T.v_to_u(v), T.u_to_v(u), T.h(h), T.v_to_u(vh), T.u_to_v(uh))
    ub, vb, _ = _turn_back(T, ru, rv, T.h(h))
    su, sv = H.interior(d, "u"), H.interior(d, "v")
    scale = max(np.abs(CAu).max(), np.abs(CAv).max())
    assert scale > 0
    assert np.abs(ub[(Ellipsis,) + su] - CAu[(Ellipsis,) + su]).max() <= 4e-15 * scale
    assert np.abs(vb[(Ellipsis,) + sv] - CAv[(Ellipsis,) + sv]).max() <= 4e-15 * scale


def test_arakawa_lamb_blend_limits(orc):
    """ARAKAWA_LAMB_BLEND (:543-588): with CORIOLIS_BLEND_F_EFF_MAX <= 2 the weights are Sadourny's energy scheme everywhere
    (:547-548), with a huge one they are Arakawa & Lamb's wherever the neighbouring Ih_q are within that factor; on a resting
    ocean of uniform thickness over a flat bottom q is uniform along x and every scheme gives the Sadourny acceleration."""
    d, M, h, u, v, uh, vh = _coriolis_case()
    su, sv = H.interior(d, "u"), H.interior(d, "v")

    def close(a, b, tol):
        for x, y, sl in ((a[0], b[0], su), (a[1], b[1], sv)):
            assert np.abs(x[(Ellipsis,) + sl] - y[(Ellipsis,) + sl]).max() <= tol * np.abs(y).max()
    sad = _corad(orc, d, M, dict(), u, v, h, uh, vh)
    close(_corad(orc, d, M, dict(Coriolis_Scheme=abi.AL_BLEND, F_eff_max_blend=2.0), u, v, h, uh, vh), sad, 1e-14)
    al = _corad(orc, d, M, dict(Coriolis_Scheme=abi.ARAKAWA_LAMB81), u, v, h, uh, vh)
    close(_corad(orc, d, M, dict(Coriolis_Scheme=abi.AL_BLEND, F_eff_max_blend=1.0e30), u, v, h, uh, vh), al, 1e-14)
    # the schemes differ on this state (the test above is not vacuous)
    assert np.abs(al[0] - sad[0]).max() > 1e-6 * np.abs(sad[0]).max()
    blend = _corad(orc, d, M, dict(Coriolis_Scheme=abi.AL_BLEND), u, v, h, uh, vh)
    assert _blend_weight_census(d, M, h) == set(range(7))
    assert np.abs(blend[0] - sad[0]).max() > 1e-6 * np.abs(sad[0]).max() and np.abs(blend[0] - al[0]).max() > 1e-6 * np.abs(sad[0]).max()


def _rk2_run(orc, d, M, first_direction, u, v, h, taux, tauy, coefs, nsteps=2, dt=900.0, bt_mod=None, sum_order=None, vv=None, hv=None):
    GV = abi.vgrid_default(); Rlay, gp = abi.layer_densities(d.nk)
    bt = abi.barotropic_params_default(30.0); bt.strong_drag = 1     # (no libm pow on the path: everything bit-comparable)
    for k_, v_ in (bt_mod or {}).items():
        setattr(bt, k_, v_)
    cont = abi.continuity_params_default(d.nk, GV.Angstrom_H)
    if sum_order is not None:
        cont.sum_order = sum_order
    m = orc.OrcModel(d, M, GV, cont, bt, abi.coriolis_params_default(), abi.pgf_params_default(), abi.rk2_params_default(), Rlay, gp,
                     first_direction)
    if vv is not None:
        m.set_vertvisc(*vv)
    if hv is not None:
        m.set_hor_visc(hv)
    s = dict(u=u.copy(), v=v.copy(), h=h.copy(), uh=np.zeros_like(h), vh=np.zeros_like(h), uhtr=np.zeros_like(h), vhtr=np.zeros_like(h),
             eta_av=np.zeros(d.shape2()))
    m.initialize(s["u"], s["v"], s["h"], s["uh"], s["vh"], dt)
    for n in range(nsteps):
        m.step(s["u"], s["v"], s["h"], s["uh"], s["vh"], s["uhtr"], s["vhtr"], s["eta_av"], taux, tauy, dt, coefs, calc_dtbt=(n == 0))
    return s, m


def _coefs(d, M, h):
    a = np.zeros((d.nk + 1,) + d.shape2()); a[1:] = 1e-5; a[d.nk] = 3e-4
    hu = np.maximum(h, 1e-9)
    return a, hu


@pytest.mark.parametrize("first_direction", [0, 1])
def test_rotate_whole_baroclinic_step(orc, first_direction):
    """Two steps of step_MOM_dyn_split_RK2 (continuity x3, CorAdCalc, PressureForce, btstep with its sub-cycle, vertvisc,
    the RK2 glue) on a closed basin and on its quarter turn: u, v, h, uh, vh, uhtr, vhtr, eta_av agree bit for bit."""
    d, M, h, u, v = _state(H.benchmark_small(), u_max=0.05, h_pert=0.001)
    T = Turn(d); dr = T.dr; Mr = T.metrics(M)
    taux = np.ascontiguousarray(0.1 * synth.smooth_field(d, 41, ox=1, oy=.5) * M[G["mask2dCu"]])
    tauy = np.ascontiguousarray(0.05 * synth.smooth_field(d, 42, ox=.5, oy=1) * M[G["mask2dCv"]])
    a, hu = _coefs(d, M, h)
    c0 = tuple(np.ascontiguousarray(x) if x is not None else None for x in
               (a * M[G["mask2dCu"]][None], a * M[G["mask2dCv"]][None], hu, hu.copy(), None, None))
    # the coefficients live on faces: a turned u face carries what the v face it came from carried
    c1 = (T.v_to_u(c0[1], 1.0), T.u_to_v(c0[0], 1.0), T.v_to_u(c0[3], 1.0), T.u_to_v(c0[2], 1.0), None, None)
    s, _ = _rk2_run(orc, d, M, first_direction, u, v, h, taux, tauy, c0)
    sr, _ = _rk2_run(orc, dr, Mr, 1 - first_direction, T.v_to_u(v), T.u_to_v(u), T.h(h), T.v_to_u(tauy), T.u_to_v(taux), c1)
    su, sv, sh = H.interior(d, "u"), H.interior(d, "v"), H.interior(d, "h")
    for a_, b_ in (("u", "v"), ("uh", "vh"), ("uhtr", "vhtr")):
        ub, vb, hb = _turn_back(T, sr[a_], sr[b_], sr["h"])
        H.assert_bitwise(ub, s[a_], "rotate:" + a_, su); H.assert_bitwise(vb, s[b_], "rotate:" + b_, sv)
    H.assert_bitwise(hb, s["h"], "rotate:h", sh)
    H.assert_bitwise(_turn_back(T, sr["u"], sr["v"], sr["eta_av"])[2], s["eta_av"], "rotate:eta_av", sh)
    assert np.abs(s["u"]).max() > 1e-3


HV_ROT = {
    "biharmonic": dict(Ah_vel_scale=0.01),
    "laplacian": dict(Laplacian=1, biharmonic=0, Kh=500.0, Kh_vel_scale=0.02),
    "smagorinsky_both_better_bounds": dict(Laplacian=1, Smagorinsky_Kh=1, Smag_Lap_const=0.15, Smagorinsky_Ah=1, Smag_bi_const=0.06, Kh=10.0,
                                           Ah=1.0e8),
    "smagorinsky_bound_coriolis_legacy": dict(Smagorinsky_Ah=1, Smag_bi_const=0.06, bound_Coriolis=1, bound_Cor_vel=2.0, Ah=1.0e8,
                                              better_bound_Ah=0),
    "noslip_laplacian": dict(Laplacian=1, biharmonic=0, Kh=800.0, no_slip=1),
    # Leith: on a grid with UNIFORM metrics only (see the test)
    "leith_kh@cartesian": dict(Laplacian=1, Leith_Kh=1, Leith_Lap_const=1.0, Kh=10.0, Ah=1.0e8),
    "leith_kh_beta_modified_les@cartesian": dict(Laplacian=1, Leith_Kh=1, Leith_Lap_const=1.5, use_beta_in_Leith=1, modified_Leith=1,
                                                 add_LES_viscosity=1, Kh=10.0, Ah=1.0e8),
    "leith_ah@cartesian": dict(Leith_Ah=1, Leith_bi_const=5.0, Ah=1.0e7),
}


@pytest.mark.parametrize("flags", sorted(HV_ROT))
def test_rotate_horizontal_viscosity(orc, flags):
    """horizontal_viscosity (+ hor_visc_init's 2-D planes) on a closed basin with an island and on its quarter turn: the
    accelerations of the turned run, turned back (diffu = diffv', diffv = -diffu'), equal the original ones bit for bit --
    Laplacian, biharmonic, Smagorinsky, the stability bounds, NOSLIP, and the Leith viscosity from the vorticity gradient
    (with beta and the divergence gradient).  The Leith gradients are turn-symmetric on uniform metrics only: the
    reference scales the x-derivative of the vorticity at a v point with DY_dxBu of the vertex to its EAST and the y-derivative
    at a u point with DX_dyBu of the vertex to its NORTH (MOM_hor_visc.F90:990-998), and a quarter turn takes north to west;
    on the sphere the two readings differ by the metric's variation over one cell (1e-3 here) -- in the reference itself.
    So the Leith cases run on a Cartesian beta-plane basin, where they pin the index conventions bit for bit."""
    if flags.endswith("@cartesian"):
        from mom6_amd import grid
        gg = grid.GlobalGrid(30, 22, kind="cartesian", dx=2.0e4, dy=2.0e4, f0=1.0e-4, beta=2.0e-11, depth_fn=grid.bowl_depth(30, 22, 3000.0))
        d, M = gg.tile(4)
    else:
        gg, d, M = H.island_basin()
    GV = abi.vgrid_default()
    h, u, v = synth.make_state(d, M, thin_frac=0.15)
    T = Turn(d); dr = T.dr; Mr = T.metrics(M)
    P = abi.hor_visc_params_default(1200.0)
    for k_, v_ in HV_ROT[flags].items():
        setattr(P, k_, v_)
    du, dv = np.zeros_like(u), np.zeros_like(v)
    orc.horizontal_viscosity(d, M, GV, P, orc.hor_visc_init(d, M, P), u, v, h, du, dv)
    ur, vr, hr = T.v_to_u(v), T.u_to_v(u), T.h(h)
    dur, dvr = np.zeros_like(ur), np.zeros_like(vr)
    orc.horizontal_viscosity(dr, Mr, GV, P, orc.hor_visc_init(dr, Mr, P), ur, vr, hr, dur, dvr)
    ub, vb, _ = _turn_back(T, dur, dvr, hr)
    H.assert_bitwise(ub, du, "rotate:diffu", H.interior(d, "u")); H.assert_bitwise(vb, dv, "rotate:diffv", H.interior(d, "v"))
    assert np.abs(du).max() > 0 and np.abs(dv).max() > 0


def test_rotate_whole_step_with_vertvisc_coef_and_horizontal_viscosity(orc):
    """The step with every callee inside it -- vertvisc_coef (x3, from visc%Kv_bbl / bbl_thick / Kv_shear) and
    horizontal_viscosity (Laplacian + biharmonic + Smagorinsky) -- on an island basin and on its quarter turn, two steps."""
    d, M, h, u, v = _state(H.island_basin(), u_max=0.05, h_pert=0.001)
    T = Turn(d); dr = T.dr; Mr = T.metrics(M)
    taux = np.ascontiguousarray(0.1 * synth.smooth_field(d, 41, ox=1, oy=.5) * M[G["mask2dCu"]])
    tauy = np.ascontiguousarray(0.05 * synth.smooth_field(d, 42, ox=.5, oy=1) * M[G["mask2dCv"]])
    a, hu = _coefs(d, M, h)
    c0 = tuple(np.ascontiguousarray(x) if x is not None else None for x in
               (a * M[G["mask2dCu"]][None], a * M[G["mask2dCv"]][None], hu, hu.copy(), None, None))
    c1 = (T.v_to_u(c0[1], 1.0), T.u_to_v(c0[0], 1.0), T.v_to_u(c0[3], 1.0), T.u_to_v(c0[2], 1.0), None, None)
    P = abi.vertvisc_params_default()
    kbu = np.ascontiguousarray((2e-3 * (1 + 0.5 * synth.smooth_field(d, 91, ox=1, oy=.5))) * M[G["mask2dCu"]])
    kbv = np.ascontiguousarray((2e-3 * (1 + 0.5 * synth.smooth_field(d, 92, ox=.5, oy=1))) * M[G["mask2dCv"]])
    btu = np.ascontiguousarray(8.0 * (1 + 0.6 * synth.smooth_field(d, 93, ox=1, oy=.5)))
    btv = np.ascontiguousarray(8.0 * (1 + 0.6 * synth.smooth_field(d, 94, ox=.5, oy=1)))
    ksh = np.ascontiguousarray(1e-3 * np.abs(synth.smooth_field(d, 95, nk=d.nk + 1, ox=.5, oy=.5)))
    vv0 = (P, kbu, kbv, btu, btv, ksh, None, None)
    vv1 = (P, T.v_to_u(kbv, 1.0), T.u_to_v(kbu, 1.0), T.v_to_u(btv, 1.0), T.u_to_v(btu, 1.0), T.h(ksh), None, None)
    hv = abi.hor_visc_params_default(900.0)
    hv.Laplacian = 1; hv.Kh = 200.0; hv.Smagorinsky_Kh = 1; hv.Smag_Lap_const = 0.15; hv.Smagorinsky_Ah = 1; hv.Smag_bi_const = 0.06
    hv.Ah_vel_scale = 0.02
    s, _ = _rk2_run(orc, d, M, 0, u, v, h, taux, tauy, c0, vv=vv0, hv=hv)
    sr, _ = _rk2_run(orc, dr, Mr, 1, T.v_to_u(v), T.u_to_v(u), T.h(h), T.v_to_u(tauy), T.u_to_v(taux), c1, vv=vv1, hv=hv)
    su, sv, sh = H.interior(d, "u"), H.interior(d, "v"), H.interior(d, "h")
    for a_, b_ in (("u", "v"), ("uh", "vh"), ("uhtr", "vhtr")):
        ub, vb, hb = _turn_back(T, sr[a_], sr[b_], sr["h"])
        H.assert_bitwise(ub, s[a_], "rotate:" + a_, su); H.assert_bitwise(vb, s[b_], "rotate:" + b_, sv)
    H.assert_bitwise(hb, s["h"], "rotate:h", sh)
    assert np.abs(s["u"]).max() > 1e-3


@pytest.mark.parametrize("dim", ["t", "l", "h", "z", "r"])
def test_dim_rescaling_of_the_whole_step_is_bit_identical(orc, dim):
    """.testing dim.t / dim.l / dim.h / dim.z / dim.r: the units of time, horizontal length, thickness, depth or density are
    scaled by 2**11 (every input, metric and dimensional parameter with its own power), two baroclinic steps are taken,
    and the unscaled answers equal the unscaled run's bit for bit."""
    p = 11
    d, M, h, u, v = _state(H.benchmark_small(), u_max=0.05, h_pert=0.001)
    s = 2.0 ** p
    T_, L_, H_, Z_, R_ = (s if dim == c else 1.0 for c in "tlhzr")
    GV = abi.vgrid_default(); Rlay, gp = abi.layer_densities(d.nk)
    dt = 900.0
    taux = np.ascontiguousarray(0.1 * synth.smooth_field(d, 41, ox=1, oy=.5) * M[G["mask2dCu"]])
    tauy = np.ascontiguousarray(0.05 * synth.smooth_field(d, 42, ox=.5, oy=1) * M[G["mask2dCv"]])
    a, hu = _coefs(d, M, h)

    def run(T, L, Hs, Z, R):
        M2 = M.copy()
        for n in abi.METRICS:
            if n.startswith(("dx", "dy")): M2[G[n]] = M[G[n]] * L
            elif n.startswith(("Idx", "Idy")): M2[G[n]] = M[G[n]] / L
            elif n.startswith("area"): M2[G[n]] = M[G[n]] * L * L
            elif n.startswith("Iarea"): M2[G[n]] = M[G[n]] / (L * L)
            elif n == "bathyT": M2[G[n]] = M[G[n]] * Z
            elif n == "CoriolisBu": M2[G[n]] = M[G[n]] / T
            elif n == "Coriolis2Bu": M2[G[n]] = M[G[n]] / (T * T)
        GV2 = abi.vgrid_default()
        GV2.g_Earth = GV.g_Earth * L * L / (Z * T * T)      # [L2 Z-1 T-2]
        GV2.Rho0 = GV.Rho0 * R
        GV2.Angstrom_H = GV.Angstrom_H * Hs; GV2.H_subroundoff = GV.H_subroundoff * Hs; GV2.dZ_subroundoff = GV.dZ_subroundoff * Z
        GV2.H_to_Z = GV.H_to_Z * Z / Hs; GV2.Z_to_H = GV.Z_to_H * Hs / Z
        GV2.H_to_RZ = GV.H_to_RZ * R * Z / Hs; GV2.RZ_to_H = GV.RZ_to_H * Hs / (R * Z)
        bt = abi.barotropic_params_default(30.0); bt.strong_drag = 1
        cont = abi.continuity_params_default(d.nk, GV2.Angstrom_H)
        cont.tol_vel = cont.tol_vel * L / T
        pgf = abi.pgf_params_default(GV2.Rho0)
        Rl2 = np.ascontiguousarray(Rlay * R); gp2 = np.ascontiguousarray(gp * L * L / (Z * T * T))
        m = orc.OrcModel(d, np.ascontiguousarray(M2), GV2, cont, bt, abi.coriolis_params_default(), pgf, abi.rk2_params_default(), Rl2, gp2, 0)
        st = dict(u=u * L / T, v=v * L / T, h=h * Hs, uh=np.zeros_like(h), vh=np.zeros_like(h), uhtr=np.zeros_like(h), vhtr=np.zeros_like(h),
                  eta_av=np.zeros(d.shape2()))
        st = {k: np.ascontiguousarray(x) for k, x in st.items()}
        # coupling coefficients a_u [H T-1], thicknesses h_u [H]; wind stress [R Z L T-2]
        c = tuple(np.ascontiguousarray(x) for x in (a * Hs / T * M[G["mask2dCu"]][None], a * Hs / T * M[G["mask2dCv"]][None], hu * Hs, hu * Hs)) + (None, None)
        tx = np.ascontiguousarray(taux * R * Z * L / (T * T)); ty = np.ascontiguousarray(tauy * R * Z * L / (T * T))
        m.initialize(st["u"], st["v"], st["h"], st["uh"], st["vh"], dt * T)
        for n in range(2):
            m.step(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"], tx, ty, dt * T, c, calc_dtbt=(n == 0))
        return dict(u=st["u"] * T / L, v=st["v"] * T / L, h=st["h"] / Hs, uh=st["uh"] * T / (Hs * L * L), vh=st["vh"] * T / (Hs * L * L),
                    uhtr=st["uhtr"] / (Hs * L * L), vhtr=st["vhtr"] / (Hs * L * L), eta_av=st["eta_av"] / Hs)

    ref = run(1.0, 1.0, 1.0, 1.0, 1.0)
    out = run(T_, L_, H_, Z_, R_)
    stag = dict(u="u", v="v", h="h", uh="u", vh="v", uhtr="u", vhtr="v", eta_av="h")
    for n in ref:
        H.assert_bitwise(out[n], ref[n], f"dim.{dim}:{n}", H.interior(d, stag[n]))
    assert np.abs(ref["u"]).max() > 1e-3


# ---- tracer advection ------------------------------------------------------------------------------------------------
def _tracer_case(orc, cfg, scale=3.0):
    """A consistent (h_start, h_end, uhtr, vhtr) set: one continuity_PPM step's transports accumulated over dt."""
    gg, d, M = cfg
    GV = abi.vgrid_default(); CS = abi.continuity_params_default(d.nk, GV.Angstrom_H)
    h, u, v = synth.make_state(d, M, thin_frac=0.05)
    dt = 1800.0
    hn = np.zeros_like(h); uh = np.zeros_like(h); vh = np.zeros_like(h)
    orc.continuity_PPM(d, M, GV, CS, 0, np.ascontiguousarray(u * scale), np.ascontiguousarray(v * scale), h, hn, uh, vh, dt)
    return d, M, GV, h, hn, np.ascontiguousarray(uh * dt), np.ascontiguousarray(vh * dt), dt


@pytest.mark.parametrize("scheme", [0, 1, 2])   # PLM, PPM:H3, PPM
def test_advect_tracer_conserves_and_stays_in_bounds(orc, scheme):
    """advect_tracer (MOM_tracer_advect.F90:53): the tracer content sum(Tr h areaT) of a closed basin is conserved to
    round-off, the advected tracer stays inside the bounds of the initial one (the schemes are monotone), a uniform
    tracer stays uniform to round-off, and the remaining transports are zero when the iteration has finished."""
    d, M, GV, h0, h1, uhtr, vhtr, dt = _tracer_case(orc, H.benchmark_small())
    sl = H.interior(d, "h"); A = M[G["areaT"]][sl]; wet = (M[G["mask2dT"]][sl] > 0)
    tr = np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 71, nk=d.nk, ox=0.5, oy=0.5) * M[G["mask2dT"]][None])
    one = np.ascontiguousarray(np.full_like(h0, 3.25))
    lo, hi = tr[(Ellipsis,) + sl][:, wet].min(), tr[(Ellipsis,) + sl][:, wet].max()
    c0 = (tr[(Ellipsis,) + sl] * h0[(Ellipsis,) + sl] * A).sum()
    uhr, vhr = np.zeros_like(h0), np.zeros_like(h0)
    it = orc.advect_tracer(d, M, GV, 0, dt, scheme, h1, uhtr, vhtr, dt, [tr, one], uhr_out=uhr, vhr_out=vhr)
    assert it >= 1
    c1 = (tr[(Ellipsis,) + sl] * h1[(Ellipsis,) + sl] * A).sum()
    assert abs(c1 / c0 - 1.0) < 5e-15
    t = tr[(Ellipsis,) + sl][:, wet]
    assert t.min() >= lo - 1e-12 and t.max() <= hi + 1e-12
    assert np.abs(one[(Ellipsis,) + sl][:, wet] - 3.25).max() < 1e-14
    assert np.abs(uhr).max() == 0.0 and np.abs(vhr).max() == 0.0


def test_advect_tracer_rotate(orc):
    """advect_tracer on a grid and on its quarter turn (x_first flipped): the same tracer bit for bit."""
    d, M, GV, h0, h1, uhtr, vhtr, dt = _tracer_case(orc, H.benchmark_small())
    T = Turn(d); dr = T.dr; Mr = T.metrics(M)
    tr = np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 71, nk=d.nk, ox=0.5, oy=0.5) * M[G["mask2dT"]][None])
    trr = T.h(tr)
    for scheme in (0, 2):
        a, b = tr.copy(), trr.copy()
        orc.advect_tracer(d, M, GV, 0, dt, scheme, h1, uhtr, vhtr, dt, [a])
        orc.advect_tracer(dr, Mr, GV, 1, dt, scheme, T.h(h1), T.v_to_u(vhtr), T.u_to_v(uhtr), dt, [b])
        H.assert_bitwise(b, T.h(a), "rotate:tracer", H.interior(dr, "h"))


# ---- barotropic solver -----------------------------------------------------------------------------------------------
def test_btstep_free_surface_budget(orc):
    """btstep (MOM_barotropic.F90:455): over a baroclinic step the free surface changes by the divergence of the
    time-averaged barotropic transports it returns plus the mass source it spreads over the step,
        eta_out - eta_in = dt * (eta_src / dtbt - IareaT * div(uhbtav, vhbtav))      (cf. :2721-2727, :1549-1587)
    -- checked through its area integral on a closed basin (the transports then drop out exactly: what is left is the
    correction source), and eta_out differs from the transport-implied eta by the filter weights only."""
    gg, d, M = H.benchmark_small()
    GV = abi.vgrid_default(); Rlay, gp = abi.layer_densities(d.nk)
    h, u, v = synth.make_state(d, M, u_max=0.05, h_pert=0.001)
    dt = 900.0
    taux = np.ascontiguousarray(0.1 * synth.smooth_field(d, 41, ox=1, oy=.5) * M[G["mask2dCu"]]); tauy = np.zeros(d.shape2())
    a, hu = _coefs(d, M, h)
    c0 = tuple(np.ascontiguousarray(x) if x is not None else None for x in
               (a * M[G["mask2dCu"]][None], a * M[G["mask2dCv"]][None], hu, hu.copy(), None, None))
    s, m = _rk2_run(orc, d, M, 0, u, v, h, taux, tauy, c0, nsteps=2, dt=dt)
    sl = H.interior(d, "h"); A = M[G["areaT"]][sl]
    # (1) the layer volume equals the barotropic one: sum_k h - bathyT == eta to within the correction the next step applies
    eta_h = (s["h"].sum(0) - M[G["bathyT"]])[sl]
    assert np.abs(eta_h - m["eta"][sl]).max() < 1e-6 * np.abs(m["eta"][sl]).max() + 1e-9
    # (2) the divergence of the time-mean transports integrates to zero over the closed basin ...
    uhbt, vhbt = m["uhbt"], m["vhbt"]
    st = d.pitch
    div = (uhbt[sl] - np.roll(uhbt, 1, 1)[sl]) + (vhbt[sl] - np.roll(vhbt, 1, 0)[sl])
    assert abs(div.sum()) < 1e-9 * np.abs(uhbt).max()
    # ... and it is what changed the column volumes: sum_k (h_new - h_old) areaT = -dt div(sum_k uh) with sum_k uh = uhbt to ETA_TOLERANCE
    su = H.interior(d, "u")
    err = np.abs((s["uh"].sum(0) - uhbt)[su]) * dt * M[G["IareaT"]][su]
    assert err.max() < 1e-6


@pytest.mark.parametrize("reservoir", [False, True])
def test_tracer_vertdiff_sinking_conserves_the_tracer(orc, reservoir):
    """tracer_vertdiff with sink_rate (MOM_tracer_diabatic.F90:123-179), restated in oracle/orc_tracer.c: sinking and mixing move
    tracer between the layers of a column and, with btm_reservoir, into the reservoir -- nothing else.  sum_k (h + h_neglect) tr
    (+ the reservoir's gain in H units) is unchanged to round-off; without the reservoir the limited sinking distances never let
    anything through the bottom."""
    from mom6_amd import abi, synth
    from tests import helpers as H
    gg, d, M = H.benchmark_small(nk=12)
    GV = abi.vgrid_default()
    nk = d.nk
    h, _, _ = synth.make_state(d, M, thin_frac=0.15)
    ent = np.abs(synth.smooth_field(d, 81, nk=nk + 1, ox=0.5, oy=0.5)) * 5.0
    ent[0] = 0.0; ent[nk] = 0.0
    ea = np.ascontiguousarray(ent[:nk]); eb = np.ascontiguousarray(ent[1:])
    T0 = np.ascontiguousarray(10.0 + 5.0 * synth.smooth_field(d, 82, nk=nk, ox=0.5, oy=0.5))
    dt = 3600.0
    sl = (Ellipsis,) + tuple(H.interior(d, "h"))
    wet = M[abi.G["mask2dT"]][sl[1:]] > 0
    sink_rate = 0.7 * float(h[sl].mean()) / dt
    res = np.zeros(d.shape2()) if reservoir else None
    T = T0.copy()
    orc.tracer_vertdiff_sink(d, M, GV, h, ea, eb, dt, T, sink_rate, None, None, res, True)
    hh = h[sl] + GV.H_subroundoff
    before = (hh * T0[sl]).sum(0); after = (hh * T[sl]).sum(0)
    gain = res[sl[1:]] * GV.RZ_to_H if reservoir else 0.0
    err = np.abs(after + gain - before)[wet] / np.abs(before)[wet]
    assert err.max() < 1e-13, err.max()
    if reservoir:
        assert (res[sl[1:]][wet] > 0).all()
    else:     # nothing leaves through the bottom: the tracer piles up there (sinking relative to the water)
        assert (T[sl][-1][wet] > T0[sl][-1][wet]).mean() > 0.9
