"""Registers and scratch memory of every gfx950 kernel as the compiler reports them (`-Rpass-analysis=kernel-resource-usage`,
recorded by mom6_amd/build.py in mom6_amd/lib/kernel_resources.json).  No kernel may keep values in scratch memory unless it is on
the list below with a reason: a device function shared between kernels that grows can push a hot kernel over its register budget
without any parity test noticing (round 3: three Coriolis schemes added to `corad_acc_layer` cost `k_corad_fused`, which never
executes them, 14 spilled registers and 1.6 ms per call; `k_hv_fused` was shelved as "slower than four kernels" while 56 of its
values lived in scratch)."""
import json
import os
import re

from mom6_amd import build

# kernel-name pattern -> (most bytes of scratch per lane tolerated, why)
ALLOWED = {
    r"k_mass_flux_waveILi[01]ELi\dELb1E": (400, "the STATS instantiation (mom6x_continuity_stats): runs for one step after bench.py's timed region"),
    r"k_corad_fused": (16, "2 of 128 registers at 4 wavefronts per SIMD; at 136 registers only one 512-thread work-group fits a CU"),
}
# the kernels the step spends its time in: named, so that a rename cannot silently drop them from the check
HOT = ["k_mass_flux_waveILi0ELi5ELb0E", "k_mass_flux_waveILi1ELi5ELb0E", "k_corad_fused", "k_hv_fusedILi32ELi24E", "k_vertvisc_coefILi0ELi3E",
       "k_vertvisc_colsILi0ELb0ELb1ELi75E", "k_vertvisc_coef_colsILi0ELi3ELb1ELb1ELi75E", "k_vertvisc_coef_colsILi0ELi1ELb1ELb0ELi75E", "k_bt_velILi0E", "k_bt_colILi0E", "k_convergenceILi0E", "k_pgf_main", "k_ta_x_tileILi4E",
       "k_ta_y_tileILi4ELi32E", "k_tridiag_colsILi75ELb0E", "k_remap_apply", "k_remap_merge", "k_remap_recon"]


def _resources():
    build.build()
    path = os.path.join(os.path.dirname(build.LIB), "kernel_resources.json")
    assert os.path.exists(path), "mom6_amd/build.py did not write kernel_resources.json"
    res = {}
    for f, ks in json.load(open(path)).items():
        for k, v in ks.items():
            res[k] = dict(v, file=f)
    return res


def test_no_kernel_spills_unless_it_is_listed_with_a_reason():
    res = _resources()
    assert len(res) > 120
    bad = []
    for k, v in res.items():
        limit = 0
        for pat, (lim, _why) in ALLOWED.items():
            if re.search(pat, k):
                limit = lim
        if v.get("scratch", 0) > limit or (limit == 0 and v.get("vgpr_spill", 0) > 0):
            bad.append((v["file"], k, v))
    assert not bad, "kernels that keep values in scratch memory:\n" + "\n".join(map(str, bad))


def test_the_hot_kernels_are_in_the_record_and_keep_their_occupancy():
    res = _resources()
    for name in HOT:
        hit = [k for k in res if name in k]
        assert hit, name
    # the dominant kernel: 2 wavefronts per SIMD and no scratch is what DESIGN.md section 4 states
    for k, v in res.items():
        if re.search(r"k_mass_flux_waveILi[01]ELi5ELb0E", k):
            assert v["scratch"] == 0 and v["occupancy"] >= 2 and v["vgprs"] <= 256, (k, v)
        if "k_hv_fusedILi32ELi24E" in k:
            assert v["scratch"] == 0 and v["occupancy"] >= 2, (k, v)
