"""MOM6's 2-D tile decomposition (MOM_domains: LAYOUT = npx,npy), one tile per GPU.

`attach_comm` wires a Dycore to its neighbours: the 128-byte RCCL unique id is created on rank 0 by the C
library, broadcast with torch.distributed (the host's job; a Fortran host would MPI_Bcast it) and handed
back to the library, which then performs every halo update of the dycore as a packed ncclSend/ncclRecv
group exchange over xGMI (mom6_amd/csrc/halo.hip).

`exchange_numpy` executes the SAME exchange plan (mom6x_halo_region / mom6x_halo_neighbor, host-only C
functions) on numpy tiles with torch.distributed point-to-point ops; it exists so that the plan can be
tested on CPU with the gloo backend (tests/test_halo_plan_cpu.py).
"""
import ctypes as C

import numpy as np

from . import abi

DIRS = ["W", "E", "S", "N", "SW", "SE", "NW", "NE"]
OPP = [1, 0, 3, 2, 7, 6, 5, 4]


def rank_to_pe(rank, layout):
    return (rank % layout[0], rank // layout[0])


def neighbor(lib, layout, pe, d, reentrant_x, reentrant_y):
    return lib.mom6x_halo_neighbor(layout[0], layout[1], pe[0], pe[1], d, int(reentrant_x), int(reentrant_y))


def region(lib, dims, stagger, d, send):
    i0, i1, j0, j1 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.mom6x_halo_region(C.byref(dims), stagger, d, int(send), C.byref(i0), C.byref(i1), C.byref(j0), C.byref(j1))
    assert rc == 0
    return i0.value, i1.value, j0.value, j1.value


def unique_id(lib):
    """A communicator id (mom6x_comm_unique_id) as 128 bytes, to be handed to every rank's attach_comm(unique_id=...)."""
    buf = C.create_string_buffer(128)
    abi.check(lib, lib.mom6x_comm_unique_id(buf))
    return buf.raw


def attach_comm(dyc, layout, pe, dist=None, force_nccl_self=False, unique_id=None):
    """Give `dyc` its place in the LAYOUT and (if more than one rank, or in test mode) an RCCL communicator.  The
    communicator id is made on rank 0 and broadcast with torch.distributed, unless the caller hands one over (ranks that
    are threads of one process: the transport of tests/transport)."""
    import torch
    lib = dyc.lib
    nranks = layout[0] * layout[1]
    need_id = nranks > 1 or force_nccl_self
    idbuf = None
    if need_id and unique_id is not None:
        idbuf = C.create_string_buffer(bytes(unique_id), 128)
    elif need_id:
        idbuf = C.create_string_buffer(128)
        rank = pe[0] + layout[0] * pe[1]
        if rank == 0:
            abi.check(lib, lib.mom6x_comm_unique_id(idbuf))
        if nranks > 1:
            assert dist is not None, "torch.distributed is needed to broadcast the RCCL unique id"
            t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).clone()
            if dist.get_backend() == "nccl":
                t = t.to(dyc.device)
            dist.broadcast(t, src=0)
            idbuf = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
    abi.check(lib, lib.mom6x_comm_init(dyc.ctx, layout[0], layout[1], pe[0], pe[1], idbuf, int(force_nccl_self)))


def pass_fields(dyc, fields, staggers):
    """do_group_pass of a list of torch fields (2-D or 3-D) with their staggerings (0 h, 1 u, 2 v, 3 q)."""
    n = len(fields)
    ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in fields])
    stg = (C.c_int * n)(*staggers)
    nks = (C.c_int * n)(*[1 if f.dim() == 2 else f.shape[0] for f in fields])
    abi.check(dyc.lib, dyc.lib.mom6x_pass_fields(dyc.ctx, ptrs, stg, nks, n))


def exchange_numpy(lib, dims, fields, staggers, layout, pe, dist):
    """The exchange plan of halo.hip executed on numpy tiles (pitched layout) over torch.distributed p2p."""
    import torch
    rank = pe[0] + layout[0] * pe[1]
    nbr = [neighbor(lib, layout, pe, d, dims.reentrant_x, dims.reentrant_y) for d in range(8)]

    def view(f, r):
        i0, i1, j0, j1 = r
        sl = dims.sl(i0, i1, j0, j1)
        return f[(Ellipsis,) + sl]

    send = {}
    for d in range(8):
        if nbr[d] < 0:
            continue
        parts = [np.ascontiguousarray(view(f, region(lib, dims, s, d, True))).ravel() for f, s in zip(fields, staggers)]
        send[d] = np.concatenate(parts)
    recv = {}
    ops, keep = [], []
    # sends in direction order, receives in the order of the opposite directions (see halo.hip)
    for d in range(8):
        if nbr[d] < 0:
            continue
        if nbr[d] == rank:
            recv[OPP[d]] = send[d].copy()
        else:
            t = torch.from_numpy(send[d]); keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, nbr[d], tag=0))
    for d in range(8):
        r = OPP[d]
        if nbr[r] < 0 or nbr[r] == rank:
            continue
        t = torch.empty(send[r].shape[0], dtype=torch.float64); keep.append(t)
        recv[r] = t
        ops.append(dist.P2POp(dist.irecv, t, nbr[r], tag=0))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for r, buf in recv.items():
        buf = buf.numpy() if hasattr(buf, "numpy") else buf
        off = 0
        for f, s in zip(fields, staggers):
            v = view(f, region(lib, dims, s, r, False))
            n = v.size
            v[...] = buf[off:off + n].reshape(v.shape)
            off += n
