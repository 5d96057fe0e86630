#!/usr/bin/env python
"""Writes the fixtures under tests/golden/ from the ORACLE (oracle/liborc.so) on the seeded cases of tests/cases.py.

These are regression fixtures of this repository's own CPU restatement -- NOT outputs of the reference Fortran
(which cannot be built in this image; DESIGN.md section 2: parity unpinned).  They pin today's oracle so that
 (a) `pytest -m "not gpu"` notices any drift of the oracle (compiler, refactoring), and
 (b) the GPU parity tests can compare the HIP path with data that is under version control.
Only computational-domain slices are stored (a few hundred kB).  Run:  python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import orc          # noqa: E402
from tests import cases         # noqa: E402
from tests import helpers as H  # noqa: E402

STAG = dict(u="u", v="v", h="h", uh="u", vh="v", uhtr="u", vhtr="v", eta_av="h", u_cor="u", v_cor="v")


def main():
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for sums in ("exact", "tree", "fma"):   # the three arithmetics of the mass-flux column sums (mom6x_continuity_params.sum_order)
        os.environ["MOM6X_SUMS"] = sums
        write_set()
    for f in sorted(os.listdir(os.path.join(ROOT, "tests", "golden"))):
        print(f, os.path.getsize(os.path.join(ROOT, "tests", "golden", f)), "bytes")


def write_set():
    tag = H.golden_tag()
    cfg = H.double_gyre()
    d = cfg[1]
    so, _ = cases.oracle_rk2(orc, cfg, cases.rk2_inputs(cfg), 3, bt_mod=dict(strong_drag=1))
    np.savez_compressed(H.golden_path("rk2_double_gyre_strong_drag_3steps" + tag),
                        **{n: so[n][(Ellipsis,) + tuple(H.interior(d, STAG[n]))] for n in so})
    cfg = H.benchmark_small()
    d = cfg[1]
    out, _, _ = cases.oracle_continuity(orc, cfg, cases.continuity_inputs(cfg))
    np.savez_compressed(H.golden_path("continuity_benchmark_small_corrector" + tag),
                        **{n: out[n][(Ellipsis,) + tuple(H.interior(d, STAG[n]))] for n in out})
    cfg = H.double_gyre()
    d = cfg[1]
    so, _, _, _ = cases.oracle_shim_case(orc, cfg)       # the case of tests/fortran_stubs/drive_shims.F90
    np.savez_compressed(H.golden_path("rk2_double_gyre_shims_4steps" + tag),
                        **{n: so[n][(Ellipsis,) + tuple(H.interior(d, STAG[n]))] for n in so})
    lines = cases.oracle_ocean_stats(orc, H.double_gyre(), 3, dict(strong_drag=1))
    with open(H.golden_path("ocean.stats.double_gyre_strong_drag_3steps" + tag, ""), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
