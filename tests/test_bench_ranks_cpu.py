"""The rendezvous half of bench.py's N > 1 path on CPU: two gloo ranks go through bench.RankEnv (barrier, maximum over the
ranks), parallel.rank_to_pe and parallel.attach_comm's unique-id broadcast -- the communicator id is made on rank 0 by the
library (stubbed here: no GPU), broadcast with torch.distributed and handed to mom6x_comm_init on every rank together with the
rank's place in the LAYOUT."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    from mom6_amd import parallel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    env = bench.RankEnv(rank, world, 0, dist)
    layout = bench.LAYOUTS[world]
    pe = parallel.rank_to_pe(rank, layout)

    class Lib:                      # the two entry points attach_comm calls, without a GPU
        def __init__(self): self.init_args = None
        def mom6x_comm_unique_id(self, buf):
            C.memmove(buf, bytes((7 * n + 3) %% 256 for n in range(128)), 128)      # what rank 0 "creates"
            return 0
        def mom6x_comm_init(self, ctx, npx, npy, px, py, idbuf, force):
            self.init_args = (npx, npy, px, py, bytes(idbuf.raw), force)
            return 0
        def mom6x_last_error(self): return b""

    class Dyc:
        lib = Lib(); ctx = None; device = torch.device("cpu")

    d = Dyc()
    parallel.attach_comm(d, layout, pe, dist)
    env.barrier()
    m = env.max(float(rank + 1), torch.device("cpu"))
    open(os.path.join(os.environ["OUT_DIR"], "rank%%d.json" %% rank), "w").write(json.dumps({"rank": rank, "pe": pe, "init": [d.lib.init_args[0], d.lib.init_args[1], d.lib.init_args[2], d.lib.init_args[3]],
                      "id_ok": d.lib.init_args[4] == bytes((7 * n + 3) %% 256 for n in range(128)), "max": m}))
    dist.destroy_process_group()
""") % ROOT


def test_two_gloo_ranks_through_bench_plumbing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", OUT_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29613", str(script)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    import json
    rows = [json.load(open(tmp_path / ("rank%d.json" % n))) for n in range(2)]
    assert len(rows) == 2
    assert rows[0]["pe"] == [0, 0] and rows[1]["pe"] == [1, 0]                    # LAYOUT 2 x 1: ranks run along i first
    assert rows[0]["init"] == [2, 1, 0, 0] and rows[1]["init"] == [2, 1, 1, 0]
    assert all(x["id_ok"] for x in rows) and all(x["max"] == 2.0 for x in rows)   # rank 1 received rank 0's id; the max met everybody
