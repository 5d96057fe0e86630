"""Dev tool: sample the GPU's shader / memory clocks and power from sysfs (hwmon) while bench.py runs the headline model, and print the
distribution.  python scripts/dev/clock_sampler.py  (on the GPU box; read-only, no privileges needed)"""
import glob, os, subprocess, sys, threading, time, collections

def find():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(hw, {})[name] = p
    return out

def label(p):
    l = p.replace("_input", "_label").replace("_average", "_label")
    try:
        return open(l).read().strip()
    except OSError:
        return os.path.basename(p)

def main():
    src = find()
    print("hwmon nodes:", {k: list(v) for k, v in src.items()})
    if not src:
        for c in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
            print(c, open(c).read())
        return
    # every card of the host is visible in sysfs; this process's GPU is the one whose power rises with the run
    samples = {hw: collections.defaultdict(list) for hw in src}
    stop = [False]
    def loop():
        while not stop[0]:
            for hw, files in src.items():
                for n, p in files.items():
                    try:
                        samples[hw][n].append(int(open(p).read()))
                    except (OSError, ValueError):
                        pass
            time.sleep(0.001)
    idle = {}
    for hw, files in src.items():
        try:
            idle[hw] = int(open(files["power1_input"]).read())
        except Exception:
            idle[hw] = 0
    t = threading.Thread(target=loop); t.start()
    env = dict(os.environ, MOM6X_BENCH_NO_PMC="1")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "60", "--warmup", "5", "--no-config4", "--no-cpu-baseline", "--no-comm-model"],
                       capture_output=True, text=True, env=env)
    stop[0] = True; t.join()
    import json
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1]); print("ms_per_step", d["ms_per_step"], d.get("box_calibration"))
    except Exception as e:
        print("bench failed", e, r.stderr[-500:])
    for hw in sorted(src):
        files = src[hw]
        print(hw, "power before the run: %.0f W" % (idle[hw] / 1e6))
        for n, v in samples[hw].items():
            if not v:
                continue
            v2 = sorted(v)
            q = lambda f: v2[int(f * (len(v2) - 1))]
            print("   %-14s %-9s n=%d min %d p10 %d p50 %d p90 %d p99 %d max %d" % (n, label(files[n]), len(v), v2[0], q(.1), q(.5), q(.9), q(.99), v2[-1]))
    for c in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk") + glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk") + glob.glob("/sys/class/drm/card*/device/power_dpm_force_performance_level"):
        try:
            print(c, open(c).read().replace("\n", " | "))
        except OSError as e:
            print(c, e)

main()
