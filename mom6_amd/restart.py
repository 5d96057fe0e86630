"""Host side of MOM_restart (src/framework/MOM_restart.F90) for the device-resident dynamical core.

The reference keeps a registry of POINTERS (register_restart_field :138-152); save_restart (:1567) writes one netCDF
variable per registered field on the axes lonh / lath / lonq / latq / Layer / Interface / Time (MOM_io.F90:430-469), in
unscaled MKS, with a `checksum` attribute (:1741-1749, written with Z16, MOM_io_infra.F90:1986); restore_state (:1798)
reads them back and verifies the checksums (RESTART_CHECKSUMS_REQUIRED, default true :2269).  Here the registered
"pointers" are device arrays; the checksums are formed on the device (mom6x_field_chksum) from the arrays as they lie
in HBM, before anything is copied, and verified there after the upload.

File format: netCDF-3 64-bit offset (scipy.io.netcdf_file version 2), the classic layout the reference's files have.
Variable set of the hot path: u, v, h (MOM.F90:3863-3904); sfc, u2, v2, CAu, CAv, diffu, diffv
(register_restarts_dyn_split_RK2, MOM_dynamics_split_RK2.F90:1261-1294); ubtav, vbtav, DTBT (register_barotropic_restarts,
MOM_barotropic.F90:6285-6304).
"""
import numpy as np

from . import parallel

_STG = {"h": 0, "u": 1, "v": 2, "q": 3}
_XAX = {"h": "lonh", "u": "lonq", "v": "lonh", "q": "lonq"}
_YAX = {"h": "lath", "u": "lath", "v": "latq", "q": "latq"}


class RestartField:
    def __init__(self, name, get, put, hor_grid, z_grid, longname, units):
        self.name, self.get, self.put, self.hor_grid, self.z_grid, self.longname, self.units = name, get, put, hor_grid, z_grid, longname, units


class MOM_restart_CS:
    """The registry.  `axes`: dict(lonh, lath, lonq, latq, Layer, Interface) of coordinate values (gridLonT ... of
    MOM_io.F90:430-447; lonq / latq have one point more than lonh / lath: symmetric memory)."""

    def __init__(self, dyc, axes, checksum_required=True):
        self.dyc, self.axes, self.checksum_required = dyc, axes, checksum_required
        self.fields = []

    # -- registration ----------------------------------------------------------------------------
    def register_restart_field(self, tensor, name, hor_grid="h", z_grid="L", longname="", units=""):
        """register_restart_field (:138): a device array; hor_grid in 'huvq', z_grid 'L', 'i' or '1'."""
        self.fields.append(RestartField(name, lambda: tensor, None, hor_grid, z_grid, longname, units))

    def register_restart_scalar(self, get, put, name, longname="", units=""):
        """A 0-d variable (register_restart_field_0d :470), e.g. DTBT: get() -> float, put(float)."""
        self.fields.append(RestartField(name, get, put, "1", "1", longname, units))

    def register_restarts_dyn_split_RK2(self):
        """register_restarts_dyn_split_RK2 (MOM_dynamics_split_RK2.F90:1210; STORE_CORIOLIS_ACCEL = True) and
        register_barotropic_restarts (MOM_barotropic.F90:6253) for the context's control structures."""
        d = self.dyc
        self.register_restart_field(d.rk2_field("eta"), "sfc", "h", "1", "Free surface Height", "m")
        self.register_restart_field(d.rk2_field("u_av"), "u2", "u", "L", "Auxiliary Zonal velocity", "m s-1")
        self.register_restart_field(d.rk2_field("v_av"), "v2", "v", "L", "Auxiliary Meridional velocity", "m s-1")
        self.register_restart_field(d.rk2_field("CAu_pred"), "CAu", "u", "L", "Zonal Coriolis and advactive acceleration", "m s-2")
        self.register_restart_field(d.rk2_field("CAv_pred"), "CAv", "v", "L", "Meridional Coriolis and advactive  acceleration", "m s-2")
        self.register_restart_field(d.rk2_field("diffu"), "diffu", "u", "L", "Zonal horizontal viscous acceleration", "m s-2")
        self.register_restart_field(d.rk2_field("diffv"), "diffv", "v", "L", "Meridional horizontal viscous acceleration", "m s-2")
        self.register_restart_field(d.barotropic_field("ubtav"), "ubtav", "u", "1", "Time mean barotropic zonal velocity", "m s-1")
        self.register_restart_field(d.barotropic_field("vbtav"), "vbtav", "v", "1", "Time mean barotropic meridional velocity", "m s-1")
        self.register_restart_scalar(d.barotropic_dtbt, d.barotropic_dtbt, "DTBT", "Barotropic timestep", "seconds")

    # -- geometry --------------------------------------------------------------------------------
    def _interior(self, hor_grid):
        dm = self.dyc.dims
        i0 = -1 if hor_grid in "uq" else 0
        j0 = -1 if hor_grid in "vq" else 0
        return dm.sl(i0, dm.ni - 1, j0, dm.nj - 1)

    def _checksum(self, f):
        """The value of MOM_restart.F90:1741-1749: get_checksum_loop_ranges (:2416) is the h-point computational
        domain whatever the staggering (SYMMETRIC_RESTART_CHECKSUMS = False)."""
        if f.hor_grid == "1":
            return int(np.array([f.get()], dtype=np.float64).view(np.int64)[0])
        return self.dyc.field_chksum(f.get())

    # -- save / restore --------------------------------------------------------------------------
    def save_restart(self, path, time_days):
        """save_restart (:1567) into one file.  Returns {name: checksum}."""
        sums = {f.name: self._checksum(f) for f in self.fields}          # on the device, before any copy
        self.dyc.sync()
        variables = []
        for f in self.fields:
            if f.hor_grid == "1":
                variables.append((f.name, np.array([f.get()]), ("Time",), f))
                continue
            a = f.get().cpu().numpy()
            a = a[(Ellipsis,) + tuple(self._interior(f.hor_grid))]
            dims = ("Time",) + (("Layer",) if f.z_grid == "L" else ("Interface",) if f.z_grid == "i" else ()) + \
                (_YAX[f.hor_grid], _XAX[f.hor_grid])
            variables.append((f.name, a[None], dims, f))
        write_restart_file(path, self.axes, time_days, [(n, a, dm, dict(long_name=f.longname, units=f.units,
                                                                        checksum="%016X" % (sums[n] % 2 ** 64)))
                                                        for n, a, dm, f in variables])
        return sums

    def restore_state(self, path):
        """restore_state (:1798): read every registered field, put it where it lives, update the halos, verify the
        checksums.  Returns the model time [days] of the file."""
        time_days, data, atts = read_restart_file(path)
        dev, stg = [], []
        for f in self.fields:
            if f.name not in data:
                raise RuntimeError("MOM_restart: Unable to find mandatory variable " + f.name + " in restart file " + str(path))
            if f.hor_grid == "1":
                f.put(float(data[f.name].ravel()[0]))
                continue
            t = f.get()
            host = t.cpu().numpy().copy()     # restore_state reads into the registered array: what lies outside the computational
            host[(Ellipsis,) + tuple(self._interior(f.hor_grid))] = data[f.name][0]   # domain keeps its value (halos beyond a closed edge)
            t.copy_(self.dyc.to_dev(host))
            dev.append(t); stg.append(_STG[f.hor_grid])
        import torch
        torch.cuda.synchronize()
        if dev:
            parallel.pass_fields(self.dyc, dev, stg)                      # the halo updates that follow restore_state
        self.dyc.sync()
        for f in self.fields:
            if "checksum" not in atts[f.name]:
                continue   # is_there_a_checksum = .false. (MOM_restart.F90:1916-1967): nothing to compare, never an error
            want = int(atts[f.name]["checksum"][:16], 16)
            got = self._checksum(f) % 2 ** 64
            if got != want and self.checksum_required:
                raise RuntimeError("MOM_restart(restore_state): Checksum of input field %s %016X does not match value %016X stored in %s"
                                   % (f.name, got, want, path))
        return time_days


# ---- the file itself (numpy only) -----------------------------------------------------------------------------------
_AXIS_ATTS = {"lath": ("Latitude", "Y"), "lonh": ("Longitude", "X"), "latq": ("Latitude", "Y"), "lonq": ("Longitude", "X"),
              "Layer": ("Layer", "Z"), "Interface": ("Interface", "Z")}


def write_restart_file(path, axes, time_days, variables, axis_units=None):
    """axes: {name: 1-D values}; variables: [(name, array with a leading Time axis of 1, dim names, attributes)]."""
    from scipy.io import netcdf_file
    axis_units = axis_units or {}
    used = []
    for _, _, dims, _ in variables:
        for dn in dims:
            if dn != "Time" and dn not in used:
                used.append(dn)
    order = [a for a in ("lath", "lonh", "latq", "lonq", "Layer", "Interface") if a in used]     # MOM_io.F90:574-579
    with netcdf_file(str(path), "w", version=2) as nc:
        nc.createDimension("Time", None)                     # the record dimension
        for a in order:
            nc.createDimension(a, len(axes[a]))
        for a in order:
            v = nc.createVariable(a, "d", (a,))
            v[:] = np.asarray(axes[a], dtype=np.float64)
            v.long_name = _AXIS_ATTS[a][0]; v.units = axis_units.get(a, "degrees" if a[0] == "l" else "meter")
            v.cartesian_axis = _AXIS_ATTS[a][1]
        tv = nc.createVariable("Time", "d", ("Time",))
        tv.long_name = "Time"; tv.units = "days"; tv.cartesian_axis = "T"
        tv[0] = time_days
        for name, arr, dims, atts in variables:
            v = nc.createVariable(name, "d", dims)
            for k, val in atts.items():
                setattr(v, k, val)
            v[0] = np.asarray(arr, dtype=np.float64)[0]


def read_restart_file(path):
    """-> (time [days], {name: array incl. the Time axis}, {name: attributes}) of the non-axis variables."""
    from scipy.io import netcdf_file
    data, atts = {}, {}
    with netcdf_file(str(path), "r", mmap=False) as nc:
        t = float(nc.variables["Time"][0])
        for name, v in nc.variables.items():
            if name in nc.dimensions:
                continue
            data[name] = np.array(v[:], dtype=np.float64)
            atts[name] = {k: (val.decode() if isinstance(val, bytes) else val) for k, val in v._attributes.items()}
    return t, data, atts


def axes_of(gg, d, sLayer=None, sInterface=None):
    """The axis values of a GlobalGrid tile: cell centres and corners (gridLonT, gridLonB ... of the reference)."""
    ig = d.i_glob0 + np.arange(d.ni); jg = d.j_glob0 + np.arange(d.nj)
    if gg.kind == "spherical":
        x0, dx, y0, dy = gg.lon0, gg.dlon, gg.lat0, gg.dlat
    else:
        x0, dx, y0, dy = 0.0, gg.dx * 1.0e-3, 0.0, gg.dy * 1.0e-3
    nk = d.nk
    return dict(lonh=x0 + dx * (ig + 0.5), lath=y0 + dy * (jg + 0.5), lonq=x0 + dx * np.arange(d.i_glob0, d.i_glob0 + d.ni + 1),
                latq=y0 + dy * np.arange(d.j_glob0, d.j_glob0 + d.nj + 1),
                Layer=np.arange(1, nk + 1, dtype=np.float64) if sLayer is None else np.asarray(sLayer),
                Interface=np.arange(1, nk + 2, dtype=np.float64) - 0.5 if sInterface is None else np.asarray(sInterface))
