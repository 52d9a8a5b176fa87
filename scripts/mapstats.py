#!/usr/bin/env python3
"""Diagnostic: sizes met by the kernels at a bench configuration (emitted / pruned components, landmarks of the
map estimate). Usage on the GPU box: python scripts/mapstats.py [steady|survey]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame

prof = sys.argv[1] if len(sys.argv) > 1 else "steady"
f = Frame(2048, 512, 64, 1002, weight_profile=prof)
p = prm3d_defaults(2048, 600, 64)
nav = navigator.PHDNavigator(p, particlecount=2048)
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.run_stages(f.z, with_alpha=True)
ne, no, J = [], [], []
for i in range(0, 2048, 64):
    cw, _, _ = nav.CorrectConditional(i)
    pw, _, _ = nav.PruneModel(i)
    ne.append(len(cw)); no.append(len(pw)); J.append(int(pw.sum()))
for name, v in (("emitted", ne), ("pruned", no), ("landmarks J", J)):
    v = np.array(v)
    print(prof, name, "min/mean/max", v.min(), v.mean(), v.max())
nav.close()

# radius of the merge ball sqrt(T^2 trace(P)) over the corrected components of particle 0, and the box they live in
nav = navigator.PHDNavigator(p, particlecount=2048)
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.run_stages(f.z, with_alpha=False)
cw, cm, cc = nav.CorrectConditional(0)
tr = np.trace(cc, axis1=1, axis2=2)
rad = np.sqrt(p.merge_threshold ** 2 * tr)
print("merge radius percentiles 1/10/50/90/99/100:", np.percentile(rad, [1, 10, 50, 90, 99, 100]))
print("box extent:", cm.max(0) - cm.min(0), "rows:", len(cw))
nav.close()
