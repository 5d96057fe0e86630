"""Dev tool: idle gaps of the compute stream in a rocprofv3 --kernel-trace CSV (which kernel ended before the gap, which began after
it, how long), summed per (before, after) pair over the traced steps.  python scripts/trace_gaps.py <kernel_trace.csv> [nsteps]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|rccl\w+|at::native::\w+|__amd_rocclr_\w+)", n)
    return m.group(1) if m else n[:40]
byq = collections.defaultdict(list)
for r in rows:
    byq[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
print("queues:", {k: len(v) for k, v in byq.items()})
main = max(byq, key=lambda k: sum(1 for e in byq[k] if e[2].startswith("k_bt_")))   # the context's compute stream
ev = sorted(byq[main])
# the timed part: the last `nsteps` steps, a step beginning with the frame copy k_h_av that follows a k_corad kernel
starts = [i for i in range(1, len(ev)) if ev[i][2].startswith("k_h_av") and ev[i - 1][2].startswith("k_corad")]
if len(starts) > int(nsteps):
    ev = ev[starts[-int(nsteps) - 1]:starts[-1]]
others = sorted(e for k, v in byq.items() if k != main for e in v if e[1] >= ev[0][0] and e[0] <= ev[-1][1])
ob = collections.Counter()
for e in others:
    ob[e[2]] += e[1] - e[0]
print("other queues during the span (ms/step):", {k: round(v / 1e6 / nsteps, 3) for k, v in ob.most_common(8)})
gaps = collections.Counter(); cnt = collections.Counter(); busy = 0
for a, b in zip(ev, ev[1:]):
    busy += a[1] - a[0]
    g = b[0] - a[1]
    if g > 0:
        gaps[(a[2], b[2])] += g; cnt[(a[2], b[2])] += 1
span = ev[-1][1] - ev[0][0]
print("main queue %s: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms (per step: %.3f idle)" % (main, len(ev), span / 1e6, busy / 1e6, (span - busy) / 1e6, (span - busy) / 1e6 / nsteps))
for (a, b), g in gaps.most_common(30):
    print("%9.3f ms/step  n/step=%5.1f avg=%6.1f us   %s -> %s" % (g / 1e6 / nsteps, cnt[(a, b)] / nsteps, g / 1e3 / cnt[(a, b)], a, b))

bk = collections.Counter(); nk_ = collections.Counter()
for e in ev:
    bk[e[2]] += e[1] - e[0]; nk_[e[2]] += 1
print("busy per kernel on the compute stream (ms/step, launches/step, avg us):")
for k, v in bk.most_common(45):
    print("  %8.3f %6.1f %8.1f  %s" % (v / 1e6 / nsteps, nk_[k] / nsteps, v / 1e3 / nk_[k], k))
