#!/usr/bin/env python3
"""Per-stream timeline of the LAST step in a rocprofv3 kernel trace (the launches after the last pause of >= 50 ms):
span, time with at least one kernel running, per-kernel sums and launch counts, idle gaps per stream.

    python3 tools/timeline.py <dir with *_kernel_trace.csv> [--gaps 0.15]
"""
import collections
import csv
import glob
import os
import re
import sys


def norm(name):
    name = name.replace("void ", "").replace("mtsv::(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)


def main():
    d = sys.argv[1]
    gap_ms = float(sys.argv[sys.argv.index("--gaps") + 1]) if "--gaps" in sys.argv else 0.15
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), norm(r["Kernel_Name"]), r.get("Stream_Id") or r.get("Queue_Id"))
            for r in csv.DictReader(open(f))]
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(r[1] for r in rows[max(0, i - 64):i]) > 50e6:
            cut = i
    rows = rows[cut:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print(f"last step: {len(rows)} launches, span {(t1 - t0) / 1e6:.2f} ms")
    ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
    depth, last, busy, hist = 0, t0, 0, collections.Counter()
    for t, dlt in ev:
        if depth > 0:
            busy += t - last
        hist[min(depth, 4)] += t - last
        last = t
        depth += dlt
    print(f"at least one kernel running: {busy / 1e6:.2f} ms; by number of kernels in flight (0,1,2,3,4+): "
          + " ".join(f"{hist[k] / 1e6:.2f}" for k in range(5)))
    per, cnt = collections.defaultdict(float), collections.Counter()
    for s, e, k, _ in rows:
        per[k] += (e - s) / 1e6
        cnt[k] += 1
    print("kernel sums (ms, launches):")
    for k in sorted(per, key=lambda k: -per[k]):
        print(f"  {k:34s} {per[k]:8.2f} {cnt[k]:5d}")
    print(f"  {'sum':34s} {sum(per.values()):8.2f}")
    streams = collections.defaultdict(list)
    for r in rows:
        streams[r[3]].append(r)
    for sid, rs in streams.items():
        idle = 0
        gaps = []
        for a, b in zip(rs, rs[1:]):
            g = (b[0] - a[1]) / 1e6
            if g > 0:
                idle += g
            if g > gap_ms:
                gaps.append((g, a[2], b[2], (a[1] - t0) / 1e6))
        print(f"stream {sid}: {len(rs)} launches, first at {(rs[0][0] - t0) / 1e6:.2f} ms, last ends {(rs[-1][1] - t0) / 1e6:.2f} ms, idle between launches {idle:.2f} ms")
        for g, a, b, at in sorted(gaps, reverse=True)[:12]:
            print(f"    gap {g:.2f} ms at {at:.2f}: {a} -> {b}")


if __name__ == "__main__":
    main()
