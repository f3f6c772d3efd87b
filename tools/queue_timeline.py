"""Per-queue occupancy of the device over the last steps of a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 10 --warmup 3 ...
    python tools/queue_timeline.py gpurun_out/kt <ms_per_step> [steps]

Answers the question the step time alone cannot: with the stream-overlapped schedule, is the iteration bound by the device
(some queue always has a kernel running) or by the host issuing launches (all queues idle)?  Reports, per HIP queue and per
step: launches, busy time, the time only this queue was running, and for the union of all queues the time nothing ran.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


MARKER, MARKS_PER_STEP = "feat_knn_pc_kernel", 8


def main():
    root, step_ms = sys.argv[1], float(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    path = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
    rows.sort()
    # window = whole periods of the iteration, cut at a kernel that runs a fixed number of times per step (the feature
    # kNN: 4 stages x 2 generator passes), so teardown after the last step cannot dilute the figures
    marks = [r[0] for r in rows if MARKER in r[3]]
    t_end = marks[-MARKS_PER_STEP]
    t0 = marks[-MARKS_PER_STEP * (steps + 1)]
    rows = [r for r in rows if t0 <= r[0] < t_end]
    span = (t_end - t0) / 1e6
    print(f"measured period {span / steps:.2f} ms/step under the tracer (bench reported {step_ms:.2f})")

    by_q = defaultdict(list)
    for s, e, q, n in rows:
        by_q[q].append((s, e, n))

    # sweep over all kernel edges: time with >=1 kernel running anywhere, and time with exactly one queue running
    ev = []
    for s, e, q, _ in rows:
        ev.append((s, 1, q))
        ev.append((e, -1, q))
    ev.sort()
    active = defaultdict(int)
    last = t0
    idle = 0
    solo = defaultdict(int)
    conc_hist = defaultdict(int)
    for t, d, q in ev:
        live = [k for k, v in active.items() if v > 0]
        dt = t - last
        if dt > 0:
            conc_hist[len(live)] += dt
            if not live:
                idle += dt
            elif len(live) == 1:
                solo[live[0]] += dt
        active[q] += d
        last = t
    print(f"trace {path}")
    print(f"window: last {steps} steps = {span:.1f} ms; all figures below are PER STEP")
    print(f"{'queue':>8} {'launches':>9} {'busy ms':>9} {'solo ms':>9} {'<30us':>7} {'median gap us':>14}")
    for q, ks in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in ks) / 1e6 / steps
        small = sum(1 for s, e, _ in ks if e - s < 30000) / steps
        gaps = sorted(max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1))
        med = gaps[len(gaps) // 2] / 1e3 if gaps else 0.0
        print(f"{q:>8} {len(ks) / steps:9.0f} {busy:9.2f} {solo[q] / 1e6 / steps:9.2f} {small:7.0f} {med:14.1f}")
    print(f"device idle (no queue running): {idle / 1e6 / steps:.2f} ms/step of {span / steps:.2f}")
    for c in sorted(conc_hist):
        print(f"  {c} queue(s) running: {conc_hist[c] / 1e6 / steps:6.2f} ms/step")

    # the main queue's gaps: how much of its idle time is short launch gaps vs long waits on other streams
    main_q = max(by_q.items(), key=lambda kv: len(kv[1]))[0]
    ks = by_q[main_q]
    gaps = [max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1)]
    bins = [(0, 2e3), (2e3, 5e3), (5e3, 10e3), (10e3, 30e3), (30e3, 100e3), (100e3, 1e12)]
    print(f"main queue {main_q}: gaps between consecutive kernels")
    for lo, hi in bins:
        sel = [g for g in gaps if lo <= g < hi]
        print(f"  {lo / 1e3:6.0f}..{hi / 1e3 if hi < 1e11 else float('inf'):6.0f} us: {len(sel) / steps:6.0f} gaps  {sum(sel) / 1e6 / steps:6.2f} ms")


if __name__ == "__main__":
    main()
