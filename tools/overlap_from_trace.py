#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and reports, per stream / queue, the busy time and how much of it overlaps kernels of
ANOTHER queue: the evidence that the two half-populations of engine.DualEngine really run beside each other.
usage: overlap_from_trace.py <kernel_trace.csv> [skip_first_seconds]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    t0 = min(int(r["Start_Timestamp"]) for r in rows) + int(skip * 1e9)
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "0")), r["Kernel_Name"]) for r in rows
          if int(r["Start_Timestamp"]) >= t0]
    queues = sorted(set(e[2] for e in ev))
    span = max(e[1] for e in ev) - min(e[0] for e in ev)
    # sweep: time with >= 1 kernel running, with kernels of >= 2 different queues running
    pts = []
    for s, e, q, _ in ev:
        pts.append((s, 1, q))
        pts.append((e, -1, q))
    pts.sort()
    active = {q: 0 for q in queues}
    last, any_t, multi_t = pts[0][0], 0, 0
    for t, d, q in pts:
        n_q = sum(1 for v in active.values() if v > 0)
        if n_q >= 1:
            any_t += t - last
        if n_q >= 2:
            multi_t += t - last
        active[q] += d
        last = t
    tower = sum(e - s for s, e, q, k in ev if "k_conv" in k)
    print("queues with kernels: %d; window %.3f s; some kernel running %.1f %% of it; kernels of two queues at once %.1f %% of it (%.1f %% of the busy time)"
          % (len(queues), span / 1e9, 100.0 * any_t / span, 100.0 * multi_t / span, 100.0 * multi_t / max(any_t, 1)))
    print("sum of kernel durations / window = %.2f (1.0 = one kernel at a time, back to back); tower kernels alone %.2f" % (
        sum(e - s for s, e, _, _ in ev) / span, tower / span))
    for q in queues:
        b = sum(e - s for s, e, qq, _ in ev if qq == q)
        print("  queue %s: %d kernels, %.1f %% of the window" % (q, sum(1 for e in ev if e[2] == q), 100.0 * b / span))


if __name__ == "__main__":
    main()
