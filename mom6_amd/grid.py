"""Synthetic horizontal grids in the pitched tile layout (host side, numpy).

The reference builds `ocean_grid_type` in src/initialization/MOM_grid_initialize.F90
(out of scope); this module is the harness that produces
the same *kind* of metric arrays analytically for the configurations of BASELINE.json
(cartesian or spherical grids, closed or zonally re-entrant, analytic bathymetry) and
cuts per-tile views of a global grid for MOM6's 2-D domain decomposition.

Masks follow initialize_masks (MOM_grid_initialize.F90:1218-1270): a T cell is ocean
when bathyT > Dmask; a face/vertex is open when every adjacent T cell is ocean;
dy_Cu = mask2dCu*dyCu, areaCu = dxCu*dy_Cu, IareaCu = mask2dCu/areaCu.
"""
import numpy as np

from .abi import G, G_COUNT, METRICS, dims_init

R_EARTH = 6.378e6
OMEGA = 7.2921e-5


def _recip(a):
    """Adcroft_reciprocal: 1/a where a != 0, else 0 (MOM_grid_initialize.F90:1275)."""
    out = np.zeros_like(a)
    nz = a != 0.0
    out[nz] = 1.0 / a[nz]
    return out


class GlobalGrid:
    """An analytic global grid of ni_glob x nj_glob T cells.

    kind = "spherical": lon0, lat0 (deg, SW corner), dlon, dlat (deg)
    kind = "cartesian": dx, dy (m), f0, beta
    depth_fn(ig, jg) -> depth [m] for global cell indices (vectorised); cells outside a
    closed boundary are land.
    """

    def __init__(self, ni_glob, nj_glob, kind="spherical", lon0=0.0, lat0=-60.0, dlon=None, dlat=None,
                 dx=1.0e4, dy=1.0e4, f0=1.0e-4, beta=0.0, reentrant_x=False, reentrant_y=False,
                 depth_fn=None, max_depth=4000.0, min_depth=0.0):
        self.ni_glob, self.nj_glob, self.kind = ni_glob, nj_glob, kind
        self.lon0, self.lat0 = lon0, lat0
        self.dlon = 360.0 / ni_glob if dlon is None else dlon
        self.dlat = 120.0 / nj_glob if dlat is None else dlat
        self.dx, self.dy, self.f0, self.beta = dx, dy, f0, beta
        self.reentrant_x, self.reentrant_y = reentrant_x, reentrant_y
        self.max_depth, self.min_depth = max_depth, min_depth
        self.depth_fn = depth_fn if depth_fn is not None else (lambda ig, jg: np.full(np.broadcast(ig, jg).shape, max_depth))

    # -- analytic geometry at (possibly half-integer) global positions ----------------------
    def _dx_dy_f(self, y_idx):
        """dx [m], dy [m] and f at global row position y_idx (cell centres are k+0.5)."""
        if self.kind == "spherical":
            lat = np.deg2rad(self.lat0 + self.dlat * y_idx)
            dx = R_EARTH * np.cos(lat) * np.deg2rad(self.dlon)
            dy = np.full_like(lat, R_EARTH * np.deg2rad(self.dlat))
            f = 2.0 * OMEGA * np.sin(lat)
        else:
            dx = np.full_like(y_idx, self.dx, dtype=np.float64)
            dy = np.full_like(y_idx, self.dy, dtype=np.float64)
            f = self.f0 + self.beta * self.dy * (y_idx - 0.5 * self.nj_glob)
        return dx, dy, f

    def depth(self, ig, jg):
        """Depth of global T cells, 0 (land) outside closed boundaries, wrapped if re-entrant."""
        ig = np.asarray(ig)
        jg = np.asarray(jg)
        inside = np.ones(np.broadcast(ig, jg).shape, dtype=bool)
        if self.reentrant_x:
            ig = np.mod(ig, self.ni_glob)
        else:
            inside &= (ig >= 0) & (ig < self.ni_glob)
        if self.reentrant_y:
            jg = np.mod(jg, self.nj_glob)
        else:
            inside &= (jg >= 0) & (jg < self.nj_glob)
        igc = np.clip(ig, 0, self.ni_glob - 1)
        jgc = np.clip(jg, 0, self.nj_glob - 1)
        dep = np.asarray(self.depth_fn(igc, jgc), dtype=np.float64)
        return np.where(inside, dep, 0.0)

    def tile(self, nk, halo=4, layout=(1, 1), pe=(0, 0)):
        """Dims + metric block of tile `pe` in a `layout` = (npx, npy) decomposition."""
        npx, npy = layout
        if self.ni_glob % npx or self.nj_glob % npy:
            raise ValueError("layout must divide the global grid")
        ni, nj = self.ni_glob // npx, self.nj_glob // npy
        d = dims_init(ni, nj, nk, halo, self.ni_glob, self.nj_glob, pe[0] * ni, pe[1] * nj,
                      self.reentrant_x, self.reentrant_y)
        return d, self.metrics(d)

    def metrics(self, d):
        nrows, P = d.nj + 2 * d.halo + 1, d.pitch
        M = np.zeros((G_COUNT, nrows, P), dtype=np.float64)
        # local index of every memory column / row
        il = np.arange(P) - d.ioff
        jl = np.arange(nrows) - d.joff
        ig = (il + d.i_glob0)[None, :]
        jg = (jl + d.j_glob0)[:, None]
        ones = np.ones((nrows, P))
        # T points at (ig+0.5, jg+0.5); Cu at (ig+1, jg+0.5); Cv at (ig+0.5, jg+1); Bu at (ig+1, jg+1)
        dxT, dyT, _ = self._dx_dy_f(jg + 0.5)
        dxCv, dyCv, fq = self._dx_dy_f(jg + 1.0)
        M[G["dxT"]] = dxT * ones; M[G["dyT"]] = dyT * ones
        M[G["dxCu"]] = dxT * ones; M[G["dyCu"]] = dyT * ones
        M[G["dxCv"]] = dxCv * ones; M[G["dyCv"]] = dyCv * ones
        M[G["dxBu"]] = dxCv * ones; M[G["dyBu"]] = dyCv * ones
        for s in ("T", "Cu", "Cv", "Bu"):
            M[G["Idx" + s]] = _recip(M[G["dx" + s]])
            M[G["Idy" + s]] = _recip(M[G["dy" + s]])
        M[G["areaT"]] = M[G["dxT"]] * M[G["dyT"]]
        M[G["IareaT"]] = _recip(M[G["areaT"]])
        M[G["areaBu"]] = M[G["dxBu"]] * M[G["dyBu"]]
        M[G["IareaBu"]] = _recip(M[G["areaBu"]])
        M[G["CoriolisBu"]] = fq * ones
        M[G["Coriolis2Bu"]] = (fq * fq) * ones

        dep = lambda di, dj: self.depth(ig + di, jg + dj)
        Dmask = self.min_depth
        bT = dep(0, 0)
        M[G["bathyT"]] = bT
        wet = lambda di, dj: dep(di, dj) > Dmask
        M[G["mask2dT"]] = wet(0, 0).astype(np.float64)
        M[G["mask2dCu"]] = (wet(0, 0) & wet(1, 0)).astype(np.float64)
        M[G["mask2dCv"]] = (wet(0, 0) & wet(0, 1)).astype(np.float64)
        M[G["mask2dBu"]] = (wet(0, 0) & wet(1, 0) & wet(0, 1) & wet(1, 1)).astype(np.float64)
        M[G["dy_Cu"]] = M[G["mask2dCu"]] * M[G["dyCu"]]
        M[G["areaCu"]] = M[G["dxCu"]] * M[G["dy_Cu"]]
        M[G["IareaCu"]] = M[G["mask2dCu"]] * _recip(M[G["areaCu"]])
        M[G["dx_Cv"]] = M[G["mask2dCv"]] * M[G["dxCv"]]
        M[G["areaCv"]] = M[G["dyCv"]] * M[G["dx_Cv"]]
        M[G["IareaCv"]] = M[G["mask2dCv"]] * _recip(M[G["areaCv"]])
        return np.ascontiguousarray(M)


def bowl_depth(ni_glob, nj_glob, max_depth=4000.0, rim=1, min_frac=0.1):
    """An analytic bowl with a `rim`-cell land border, in the spirit of
    benchmark_initialize_topography (src/user/benchmark_initialization.F90:34-75)."""
    def fn(ig, jg):
        x = (ig + 0.5) / ni_glob
        y = (jg + 0.5) / nj_glob
        D = max_depth * (min_frac + (1.0 - min_frac) * np.minimum(1.0, 16.0 * x * (1 - x) * y * (1 - y) * 1.5))
        land = (ig < rim) | (ig >= ni_glob - rim) | (jg < rim) | (jg >= nj_glob - rim)
        return np.where(land, 0.0, D)
    return fn


def flat_depth(ni_glob, nj_glob, max_depth=4000.0, rim=1, rim_x=True):
    """Flat bottom with a land rim (north/south always; east/west unless re-entrant)."""
    def fn(ig, jg):
        land = (jg < rim) | (jg >= nj_glob - rim)
        if rim_x:
            land = land | (ig < rim) | (ig >= ni_glob - rim)
        return np.where(land, 0.0, max_depth + 0.0 * ig)
    return fn


def metric_names():
    return list(METRICS)
