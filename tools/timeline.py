"""Timeline summary of a rocprofv3 kernel trace (kernel_trace.csv): how busy the device is and with what.
Usage: python tools/timeline.py <kernel_trace.csv[.gz]> [tail_fraction]
Prints, for the last `tail_fraction` (default 0.6) of the traced interval (the timed steps, after warm-up): the union of kernel
intervals (device busy time), the time-weighted number of kernels in flight, and per kernel: launches, total span, and the time
during which it was the ONLY kernel running."""
import csv
import gzip
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    op = gzip.open if path.endswith(".gz") else open
    rows = []
    with op(path, "rt") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tc2li::", "")))
    t_lo, t_hi = min(r[0] for r in rows), max(r[1] for r in rows)
    t0 = t_hi - int((t_hi - t_lo) * tail)
    marker = os.environ.get("TIMELINE_MARKER", "k_quadtree<")  # once per step: the window starts at its 3rd launch (steady state)
    marks = sorted(r[0] for r in rows if r[2].startswith(marker))
    if len(marks) >= 4:
        t0 = marks[2]
        print("window: from the 3rd launch of %s (%d steps)" % (marker, len(marks) - 2))
    rows = [r for r in rows if r[0] >= t0]
    ev = []
    for i, (a, b, _) in enumerate(rows):
        ev.append((a, 1, i)); ev.append((b, -1, i))
    ev.sort()
    live = set()
    busy = 0
    weighted = 0
    alone = defaultdict(int)
    hist = defaultdict(int)
    prev = ev[0][0]
    for t, d, i in ev:
        dt = t - prev
        if live and dt > 0:
            busy += dt
            weighted += dt * len(live)
            hist[min(len(live), 8)] += dt
            if len(live) == 1:
                alone[rows[next(iter(live))][2]] += dt
        prev = t
        if d > 0: live.add(i)
        else: live.discard(i)
    span = rows[-1][1] - t0
    tot = defaultdict(lambda: [0, 0])
    for a, b, n in rows:
        tot[n][0] += 1; tot[n][1] += b - a
    print("interval %.1f ms, device busy %.1f ms (%.1f %%), mean kernels in flight while busy %.2f" % (span / 1e6, busy / 1e6, 100.0 * busy / span, weighted / max(busy, 1)))
    print("time with k kernels in flight (ms):", {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
    print("%-44s %8s %10s %10s %10s" % ("kernel", "launches", "span ms", "avg us", "alone ms"))
    for n, (c, s) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-44s %8d %10.2f %10.1f %10.2f" % (n[:44], c, s / 1e6, s / c / 1e3, alone[n] / 1e6))


if __name__ == "__main__":
    main()
