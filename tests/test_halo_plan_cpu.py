"""The N>1 path on CPU: the halo-exchange PLAN of mom6_amd/csrc/halo.hip (regions, neighbours, send/recv
ordering -- host-only C functions of the HIP library) executed on numpy tiles by 2 gloo ranks and compared
with tiles cut from a global field.  Covers the awkward cases: a peer that is the neighbour in several
directions (2x1 re-entrant: E and W are the same rank), self-neighbours (1x2 re-entrant) and closed edges."""
import os
import socket

import numpy as np
import pytest

from mom6_amd import abi, parallel


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def truth_tile(d, stagger, nk, seed):
    """Tile of a global random field, periodic where re-entrant; NaN where the point lies beyond a closed edge."""
    xB = stagger in (1, 3); yB = stagger in (2, 3)
    rng = np.random.default_rng(seed)
    Gf = rng.standard_normal((nk, d.nj_glob + 1, d.ni_glob + 1))      # index [J+1, I+1], I = -1..ni_glob-1
    if d.reentrant_x:
        Gf[:, :, 0] = Gf[:, :, d.ni_glob]
    if d.reentrant_y:
        Gf[:, 0, :] = Gf[:, d.nj_glob, :]
    out = np.full((nk,) + d.shape2(), np.nan)
    w = d.halo
    for j in range(-w - yB, d.nj + w):
        jg = d.j_glob0 + j
        if d.reentrant_y:
            jg = jg % d.nj_glob
        elif jg < -yB or jg > d.nj_glob - 1:
            continue
        for i in range(-w - xB, d.ni + w):
            ig = d.i_glob0 + i
            if d.reentrant_x:
                ig = ig % d.ni_glob
            elif ig < -xB or ig > d.ni_glob - 1:
                continue
            out[:, j + d.joff, i + d.ioff] = Gf[:, jg + 1, ig + 1]
    return out


def _worker(rank, world, port, layout, reentrant_x, reentrant_y, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = abi.load_library()
        ni_g, nj_g, nk = 12 * layout[0], 10 * layout[1], 3
        pe = parallel.rank_to_pe(rank, layout)
        ni, nj = ni_g // layout[0], nj_g // layout[1]
        d = abi.dims_init(ni, nj, nk, 4, ni_g, nj_g, pe[0] * ni, pe[1] * nj, reentrant_x, reentrant_y)
        staggers = [0, 1, 2, 3, 0]
        nks = [nk, nk, nk, 1, 1]
        truth = [truth_tile(d, s, k, 100 + n) for n, (s, k) in enumerate(zip(staggers, nks))]
        fields = []
        for t, s in zip(truth, staggers):
            xB = s in (1, 3); yB = s in (2, 3)
            f = np.full_like(t, -999.0)
            sl = d.sl(-xB, d.ni - 1, -yB, d.nj - 1)          # the computational domain (symmetric memory)
            f[(Ellipsis,) + sl] = t[(Ellipsis,) + sl]
            fields.append(f)
        parallel.exchange_numpy(lib, d, fields, staggers, layout, pe, dist)
        bad = []
        for n, (f, t, s) in enumerate(zip(fields, truth, staggers)):
            xB = s in (1, 3); yB = s in (2, 3)
            sl = d.sl(-d.halo - xB, d.ni - 1 + d.halo, -d.halo - yB, d.nj - 1 + d.halo)
            fv, tv = f[(Ellipsis,) + sl], t[(Ellipsis,) + sl]
            inside = ~np.isnan(tv)
            if not np.array_equal(fv[inside], tv[inside]):
                bad.append(f"field {n} (stagger {s}): {np.count_nonzero(fv[inside] != tv[inside])} wrong halo points")
            if not np.all(fv[~inside] == -999.0):
                bad.append(f"field {n}: points beyond a closed edge were touched")
        q.put((rank, bad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout,rx,ry", [((2, 1), True, False), ((1, 2), True, False), ((2, 1), False, False), ((1, 2), True, True)])
def test_halo_plan_two_ranks_gloo(layout, rx, ry):
    import torch.multiprocessing as mp
    if not os.path.exists(abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, layout, rx, ry, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bad in res:
        assert not bad, f"rank {rank}: {bad}"


def test_neighbor_and_region_tables():
    lib = abi.load_library()
    # 4x2 layout of BASELINE.json configs[4], re-entrant in x: every tile has <= 8 neighbours = <= 7 distinct peers
    for py in range(2):
        for px in range(4):
            nb = [lib.mom6x_halo_neighbor(4, 2, px, py, d, 1, 0) for d in range(8)]
            assert nb[0] == (px - 1) % 4 + 4 * py and nb[1] == (px + 1) % 4 + 4 * py
            assert (nb[3] == -1) == (py == 1) and (nb[2] == -1) == (py == 0)
            assert len({n for n in nb if n >= 0}) <= 7
    d = abi.dims_init(720, 540, 75, 4)
    # u-points: symmetric memory owns I = -1..ni-1; the E neighbour needs my I = ni-5..ni-2, I receive I = ni..ni+3 from it
    assert parallel.region(lib, d, 1, 1, True) == (720 - 5, 720 - 2, 0, 539)
    assert parallel.region(lib, d, 1, 1, False) == (720, 723, 0, 539)
    assert parallel.region(lib, d, 1, 0, False) == (-5, -2, 0, 539)
    assert parallel.region(lib, d, 0, 7, True) == (716, 719, 536, 539)      # NE corner of an h-field
    assert parallel.region(lib, d, 2, 3, True) == (0, 719, 540 - 5, 540 - 2)  # v-field to the N neighbour
