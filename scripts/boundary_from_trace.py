#!/usr/bin/env python3
"""The step boundary from a rocprofv3 kernel trace of bench.py (product build): time between the end of the step's last per-particle
launch and the start of k_normalise_resample, and between its end and the start of the next step's first k_sweep; the step period.
    python scripts/boundary_from_trace.py <dir with *_kernel_trace.csv> [label]"""
import csv
import glob
import os
import sys

import numpy as np

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else d
for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "k_" in r["Kernel_Name"]]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])) for r in rows)
    big = max((g for _, _, n, g in ev if "k_sweep" in n), default=0)
    before, after, dur, period = [], [], [], []
    prev = None
    for i, (s, e, n, g) in enumerate(ev):
        if "k_normalise_resample" not in n:
            continue
        earlier = [x for x in ev[max(0, i - 10):i] if "k_alpha_density" in x[2] and x[3] == big]
        nxt = [x for x in ev[i + 1:i + 6] if "k_sweep" in x[2] and x[3] == big]
        if not earlier or not nxt:
            continue
        before.append((s - max(x[1] for x in earlier)) / 1e3)
        after.append((min(x[0] for x in nxt) - e) / 1e3)
        dur.append((e - s) / 1e3)
        if prev is not None:
            period.append((s - prev) / 1e3)
        prev = s
    if len(before) < 10:
        continue
    k = len(before) // 5   # (the first steps are cold)
    med = lambda v: float(np.median(v[k:]))
    print("%s: %d steps; last k_alpha_density end -> k_normalise_resample start %.1f us, its duration %.1f us, its end -> next k_sweep start %.1f us; "
          "boundary %.1f us of a %.1f us period (under rocprofv3)" % (label, len(before), med(before), med(dur), med(after), med(before) + med(dur) + med(after), med(period)))
