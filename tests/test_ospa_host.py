"""SURVEY row f3, product side: monorfs_amd/host/Ospa.hpp (its own assignment solver) against the oracle's restatement
of Plot.OSPA (postanalysis/Plot.cs:531-581) on random landmark sets, including the cut-off and unequal sizes."""
import os
import subprocess

import numpy as np

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_ospa_matches_the_oracle(tmp_path):
    exe = str(tmp_path / "ospa_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "ospa_check.cpp")])
    rng = np.random.default_rng(17)
    cases, text = [], []
    for t in range(40):
        na, nb = int(rng.integers(0, 14)), int(rng.integers(0, 14))
        C, P = (1.0, 1.0) if t % 3 else (float(rng.uniform(0.3, 2.0)), float(rng.choice([1.0, 2.0])))
        a = rng.uniform(-1.5, 1.5, (na, 3))
        b = (a[rng.permutation(na)][:nb] + rng.normal(0, 0.05, (min(na, nb), 3))) if t % 2 and na else rng.uniform(-1.5, 1.5, (nb, 3))
        b = np.vstack([b, rng.uniform(-1.5, 1.5, (nb - len(b), 3))]) if len(b) < nb else b
        cases.append((a, b, C, P))
        text.append("%.17g %.17g %d %d\n" % (C, P, len(a), len(b)) + "".join("%.17g %.17g %.17g\n" % tuple(x) for x in np.vstack([a, b])))
    out = subprocess.run([exe], input="".join(text), capture_output=True, text=True, check=True).stdout.split("\n")
    for (a, b, C, P), line in zip(cases, out):
        d, card = (float(v) for v in line.split())
        wd, wcard = orc.ospa(a, b, C, P)
        assert np.isclose(d, wd, rtol=1e-12, atol=1e-12) and np.isclose(card, wcard, rtol=1e-12, atol=1e-15), (len(a), len(b), d, wd)
