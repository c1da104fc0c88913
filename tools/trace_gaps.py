"""Reads a rocprofv3 --kernel-trace CSV of `bench.py` and reports, for the graph-replayed steps of the timed region, where
the wall time goes: kernel time per name, idle gaps on the main stream (by the kernel that follows the gap), concurrency
of the second stream.    python tools/trace_gaps.py <kernel_trace.csv> [steps]"""
import collections
import csv
import sys

path = sys.argv[1]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
# the timed region = the last long run of kernels; steps are delimited by the fused Adam kernel
adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[2] or "fused_adam" in r[2].lower()]
adam = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]  # the last kernel of each optimizer step
print("kernels:", len(rows), "optimizer steps:", len(adam))
if len(adam) < 6:
    sys.exit("not enough steps in the trace")
# take the longest stretch of consecutive steps with near-equal kernel counts (graph replays)
counts = [adam[i + 1] - adam[i] for i in range(len(adam) - 1)]
common = collections.Counter(counts).most_common(1)[0][0]
steps = [(adam[i], adam[i + 1]) for i in range(len(adam) - 1) if counts[i] == common]
# the graph-replayed steps of the timed region are the shortest ones (eager warm-up / per-kernel passes are longer)
span = lambda ab: rows[ab[1]][1] - rows[ab[0] + 1][0]  # noqa: E731
fastest = min(span(ab) for ab in steps)
steps = [ab for ab in steps if span(ab) <= 1.15 * fastest]
steps = steps[-min(len(steps), int(sys.argv[2]) if len(sys.argv) > 2 else 8):]
print("kernels per step:", common, "steps analysed:", len(steps))
ktime = collections.defaultdict(float)
kcount = collections.Counter()
gap_after = collections.defaultdict(float)
wall = busy = 0.0
for a, b in steps:
    seg = rows[a + 1:b + 1]
    streams = collections.Counter(r[3] for r in seg)
    main = streams.most_common(1)[0][0]
    mseg = [r for r in seg if r[3] == main]
    wall += (mseg[-1][1] - mseg[0][0]) / 1e3
    prev_end = None
    for s, e, name, st in mseg:
        short = name.split("(")[0].replace("void ", "").replace("tp3d::", "")[:60]
        ktime[short] += (e - s) / 1e3
        kcount[short] += 1
        busy += (e - s) / 1e3
        if prev_end is not None and s > prev_end:
            gap_after[short] += (s - prev_end) / 1e3
        prev_end = max(prev_end or e, e)
n = len(steps)
print("main stream per step: wall %.1f us, kernel time %.1f us, idle %.1f us" % (wall / n, busy / n, (wall - busy) / n))
print("\n%-62s %6s %9s %9s" % ("kernel", "calls", "us/step", "gap before us/step"))
for k, v in sorted(ktime.items(), key=lambda kv: -kv[1])[:45]:
    print("%-62s %6.1f %9.1f %9.1f" % (k, kcount[k] / n, v / n, gap_after[k] / n))
print("total gap %.1f us/step over %.1f launches/step" % (sum(gap_after.values()) / n, sum(kcount.values()) / n))
