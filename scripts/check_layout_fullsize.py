#!/usr/bin/env python
"""The strong-scaling runs of bench.py compared with the one-tile run on ONE GPU: the 1440 x 1080 x 75 workload of bench.py
on a 4 x 2 (8-GPU) or 2 x 2 layout, one tile per host thread (the in-process transport of tests/transport, plugged in with mom6x_comm_set_transport), against layout 1 x 1.  After the steps
the restart checksums (mom6x_field_chksum, summed over the tiles by the same all-reduce the N-GPU run uses) of every
prognostic field must be equal.  Usage: python scripts/check_layout_fullsize.py [npx npy [steps]]"""
import argparse
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                   # noqa: E402
from mom6_amd import parallel                  # noqa: E402
from mom6_amd.abi import load_library          # noqa: E402

FIELDS = ["u", "v", "h", "uh", "vh", "uhtr", "vhtr", "eta_av"]


def run(args, layout, pe, uid, out, errors):
    try:
        dyc, d, st, taux, tauy, keep = bench.build_model(args, layout, pe, 0, None, uid)
        for n in range(args.steps):
            dyc.step_MOM_dyn_split_RK2(st["u"], st["v"], st["h"], st["uh"], st["vh"], st["uhtr"], st["vhtr"], st["eta_av"], taux, tauy,
                                       args.dt, calc_dtbt=(n == 0))
        dyc.sync()
        out[pe] = {n: dyc.field_chksum(st[n]) % 2 ** 64 for n in FIELDS}
        out[pe]["dtbt"] = dyc.barotropic_dtbt()
        dyc.close()
    except Exception:                           # noqa: BLE001
        import traceback
        errors.append((pe, traceback.format_exc()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("npx", type=int, nargs="?", default=4)
    ap.add_argument("npy", type=int, nargs="?", default=2)
    ap.add_argument("steps", type=int, nargs="?", default=2)
    a = ap.parse_args()
    sz = [int(x) for x in os.environ.get("CHECK_SIZE", "1440,1080,75").split(",")]
    args = argparse.Namespace(ni=sz[0], nj=sz[1], nk=sz[2], dt=900.0, steps=a.steps)
    errors, ref, out = [], {}, {}
    run(args, (1, 1), (0, 0), None, ref, errors)
    if errors:
        raise SystemExit(errors[0][1])
    print("1 x 1:", {k: ("%016X" % v if k != "dtbt" else v) for k, v in ref[(0, 0)].items()}, flush=True)
    layout = (a.npx, a.npy)
    from tests import helpers as TH
    TH.use_threads_transport(load_library())
    uid = parallel.unique_id(load_library())
    pes = [(px, py) for py in range(layout[1]) for px in range(layout[0])]
    th = [threading.Thread(target=run, args=(args, layout, pe, uid, out, errors)) for pe in pes]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise SystemExit(errors[0][1])
    bad = [(pe, k) for pe in pes for k in ref[(0, 0)] if out[pe][k] != ref[(0, 0)][k]]
    print(f"{layout[0]} x {layout[1]}:", {k: ("%016X" % v if k != "dtbt" else v) for k, v in out[pes[0]].items()})
    if bad:
        raise SystemExit(f"MISMATCH: {bad[:8]}")
    print(f"layout {layout[0]} x {layout[1]} == layout 1 x 1 after {a.steps} steps: every field checksum and dtbt identical")


if __name__ == "__main__":
    main()
