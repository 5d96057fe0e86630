"""bench.py's own N > 1 code path on ONE GPU: `python bench.py --gpus N --transport threads` runs run_rank() -- the function the
real N-GPU job executes in every process: rank -> tile of the LAYOUT, communicator attached before the new-run initialisation,
barriers around the timed region, the maximum over the ranks, the thermodynamic step with its global iteration flags -- with the
N ranks as host threads and every halo exchange through the in-process transport, and requires the restart checksums of all
eight prognostic fields and dtbt after the timed steps to equal those of the N = 1 run.  What stays unexercised without N GPUs is
RCCL between real peers and torch.distributed's rendezvous (tests/test_bench_ranks_cpu.py covers the latter with gloo).
The second case is BASELINE.json configs[4]: the ALE cycle (PLM pressure force, PPM tracer advection, z* regridding, PPM_H4
remapping) on the 4 x 2 layout at one eighth of the 4320 x 3240 grid's edge lengths."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--transport", "threads"] + list(extra),
                       capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("gpus", [2, 4, 8])
def test_bench_main_on_a_layout_reproduces_the_one_tile_run(gpus):
    out = _bench("--gpus", str(gpus), "--ni", "360", "--nj", "272", "--steps", "4", "--warmup", "2")
    assert out["transport"] == "threads" and out["n_gpus"] == gpus and out["layout"] == {2: [2, 1], 4: [2, 2], 8: [4, 2]}[gpus]
    assert out["layout_check"]["identical"], out["layout_check"]
    assert set(out["layout_check"]["fields"]) == {"u", "v", "h", "uh", "vh", "uhtr", "vhtr", "eta_av", "dtbt"}
    assert out["config"]["thermo_steps_in_timed_region"] == 1 and "invalid_as_performance" in out


def test_config4_ale_cycle_on_the_4x2_layout():
    out = _bench("--gpus", "8", "--workload", "config4", "--ale-ni", "544", "--ale-nj", "408")
    assert out["layout"] == [4, 2] and out["layout_check"]["identical"], out["layout_check"]
    assert set(out["layout_check"]["fields"]) == {"u", "v", "h", "T", "S", "tr1", "tr2"}
