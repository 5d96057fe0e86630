"""Dev tool: oracle step time against OMP_NUM_THREADS on this host (bench.py's cpu_baseline picks the best)."""
import os, subprocess, sys
for t in (1, 8, 16, 32, 64, 128):
    env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND="close")
    out = subprocess.run([sys.executable, "-c", """
import sys; sys.path.insert(0, '.')
import bench
class A: pass
a = A(); a.ni=1440; a.nj=1080; a.nk=75; a.dt=900.0; a.cpu_steps=2
print(bench.cpu_baseline(a)['sample'])
"""], env=env, capture_output=True, text=True)
    print(t, out.stdout.strip()[-120:], out.stderr.strip()[-200:] if out.returncode else "", flush=True)
