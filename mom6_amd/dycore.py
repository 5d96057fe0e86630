"""Host-side mirror of the reference's Fortran interface for the split-RK2 hot path.

Each method has the name and argument meaning of the reference module procedure it
stands in for and forwards to the C ABI (include/mom6x.h) with DEVICE pointers taken
from torch tensors (torch is only used for HBM allocation and streams).  There is no
CPU fallback: constructing a `Dycore` without the HIP library or without a GPU raises.

Fields are torch.float64 CUDA tensors in the pitched tile layout:
  2-D: shape (nj+2*halo+1, pitch); 3-D: shape (nk, nj+2*halo+1, pitch).
"""
import ctypes as C

import numpy as np
import torch

from . import abi
from .abi import BTCont, check, load_library


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), "need contiguous float64 CUDA tensor"
    return C.c_void_p(t.data_ptr())


class BTContDev:
    """BT_cont_type (MOM_variables.F90:315-350) on the device."""

    def __init__(self, dyc, with_h=True):
        self.f = {n: dyc.zeros2() for n in BTCont._names[:12]}
        self.f["h_u"] = dyc.zeros3() if with_h else None
        self.f["h_v"] = dyc.zeros3() if with_h else None
        self.struct = BTCont()
        for n in BTCont._names:
            t = self.f[n]
            setattr(self.struct, n, t.data_ptr() if t is not None else None)

    def __getitem__(self, n):
        return self.f[n]


class Dycore:
    """One tile of the dynamical core on one MI355X."""

    def __init__(self, dims, metrics, GV=None, first_direction=0, device=0):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("mom6_amd.Dycore needs a GPU (no CPU fallback on the product path)")
        self.dims = dims
        self.GV = GV if GV is not None else abi.vgrid_default()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.metrics_host = np.ascontiguousarray(metrics, dtype=np.float64)
        assert self.metrics_host.shape == (abi.G_COUNT,) + dims.shape2()
        self.ctx = C.c_void_p()
        check(self.lib, self.lib.mom6x_ctx_create(C.byref(self.ctx), C.byref(dims), C.c_int(device),
                                                  self.metrics_host.ctypes.data_as(C.c_void_p),
                                                  C.byref(self.GV), C.c_int(first_direction)))
        self._keep = []

    def close(self):
        if self.ctx:
            self.lib.mom6x_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ------------------------------------------------------------------------------
    def zeros2(self):
        return torch.zeros(self.dims.shape2(), dtype=torch.float64, device=self.device)

    def zeros3(self, nk=None):
        return torch.zeros(self.dims.shape3(nk), dtype=torch.float64, device=self.device)

    def to_dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def sync(self):
        """Wait for the context's streams; raises on a device-side numeric error flag."""
        check(self.lib, self.lib.mom6x_ctx_sync(self.ctx))

    @property
    def stream_ptr(self):
        return self.lib.mom6x_ctx_stream(self.ctx)

    # -- MOM_continuity_PPM ------------------------------------------------------------------
    def continuity_init(self, params=None):
        """continuity_PPM_init (MOM_continuity_PPM.F90:2674)."""
        self.cont_params = params if params is not None else abi.continuity_params_default(self.dims.nk, self.GV.Angstrom_H)
        check(self.lib, self.lib.mom6x_continuity_init(self.ctx, C.byref(self.cont_params)))

    def continuity_PPM(self, u, v, hin, h, uh, vh, dt, uhbt=None, vhbt=None, visc_rem_u=None,
                       visc_rem_v=None, u_cor=None, v_cor=None, BT_cont=None, du_cor=None, dv_cor=None):
        """continuity_PPM (MOM_continuity_PPM.F90:86): optional arguments are None when absent."""
        bt = C.byref(BT_cont.struct) if BT_cont is not None else None
        check(self.lib, self.lib.mom6x_continuity_PPM(
            self.ctx, _ptr(u), _ptr(v), _ptr(hin), _ptr(h), _ptr(uh), _ptr(vh), C.c_double(dt),
            _ptr(uhbt), _ptr(vhbt), _ptr(visc_rem_u), _ptr(visc_rem_v), _ptr(u_cor), _ptr(v_cor), bt,
            _ptr(du_cor), _ptr(dv_cor)))

    continuity = continuity_PPM  # MOM_continuity.F90:6-28 pass-through name
