#!/usr/bin/env python3
"""Writes tests/golden/*.json: the known-answer vectors the reference's own NUnit tests hold for
the PHD hot path, as DATA (inputs + expected outputs).

Nothing here imports or executes the reference (it is C# and cannot run in this image). Inputs
and expected integer vectors are transcribed from the test sources; where a test builds its
expected floating-point values with a formula of its own (PHDNavigatorTest.Correct uses
Gaussian.Multiply; PHDNavigatorTest.Prune uses Gaussian.Merge) that formula is restated below
with numpy (pinv / pseudo-determinant by SVD, as the Gaussian constructor does,
Gaussian.cs:148-180), independently of oracle/ and of the product.

Sources (under /root/reference/mono-rfs-lib/Test):
  PHDNavigatorTest.cs:54-80 (setup), :85-104 PredictInitial, :106-126 PredictKnown,
                      :128-193 Correct, :195-265 Prune
  GraphCombinatoricsTest.cs:49-64 (setup), :66-172 connected components, :174-198 AssignmentValue,
                      :200-255 LinearAssignment*, :257-306 Lexicographical*, :308-355 MurtyNode children,
                      :357-404 MurtyPairing*
  SimulationTest.cs:225-270 resample
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# Config.SetLinear2DDefaults (Config.cs:214-233) + PHD constants (Config.cs:74-91)
LINEAR2D = {
    "model": "linear2d", "zdim": 2, "measurer": [6.5],   # new Linear2DMeasurer(6.5), PHDNavigatorTest.cs:62
    "R": [[5e-4, 0], [0, 5e-4]],
    "visibility_ramp": [3 * np.sqrt(5e-4), 3 * np.sqrt(5e-4)],
    "pd": 0.9, "clutter_density": 3e-7,
    "birth_covariance": [[1e-2, 0, 0], [0, 1e-2, 0], [0, 0, 1e-2]],
    "birth_weight": 0.05, "min_weight": 1e-3, "min_effective_particle": 0.1, "max_quantity": 600,
    "merge_threshold": 0.3, "exploration_threshold": 1e-5, "density_distance_threshold": 0.5,
}


def pdet(a):
    s = np.linalg.svd(np.asarray(a, float), compute_uv=False)
    s = s[s > 1e-12 * max(s.max(), 1e-300)]
    return float(np.prod(s))


class G:
    """Gaussian.cs:148-157 (constructor) and :165-180 (Canonical)."""

    def __init__(self, mean, cov, w):
        self.mean = np.asarray(mean, float)
        self.cov = np.asarray(cov, float)
        self.w = float(w)
        self.cinv = np.linalg.pinv(self.cov)
        self.det = pdet(self.cov)
        self.mult = (2 * np.pi) ** (-(len(self.mean) // 2)) / np.sqrt(self.det)   # `-mean.Length / 2`, integer division
        self.cvec = self.cinv @ self.mean

    @staticmethod
    def canonical(vec, mat, w):
        g = G.__new__(G)
        g.cvec, g.cinv = np.asarray(vec, float), np.asarray(mat, float)
        g.cov = np.linalg.pinv(g.cinv)
        g.mean = g.cov @ g.cvec
        g.det = pdet(g.cov)
        g.w = float(w)
        g.mult = (2 * np.pi) ** (-1) / np.sqrt(g.det)   # -3 / 2 == -1 in C# integer arithmetic
        return g

    def bias(self):   # CanonicalBias, Gaussian.cs:117-123
        return np.log(self.mult) - 0.5 * self.mean @ (self.cinv @ self.mean)

    def tojson(self, w=None):
        return {"w": self.w if w is None else float(w), "mean": self.mean.tolist(), "cov": self.cov.tolist()}


def multiply(a, b):   # Gaussian.Multiply, Gaussian.cs:282-288 over Fuse :253-260
    fused = G.canonical(a.cvec + b.cvec, a.cinv + b.cinv, 1.0)
    logscale = a.bias() + b.bias() - fused.bias()
    fused.w = float(np.exp(logscale + np.log(a.w) + np.log(b.w)))
    return fused


def merge(comps):   # Gaussian.Merge, Gaussian.cs:297-347
    w = 0.0
    mean = np.zeros(3)
    cov = np.zeros((3, 3))
    for c in comps:
        w += c.w
        mean = mean + c.w * c.mean
        cov = cov + c.w * (c.cov + np.outer(c.mean, c.mean))
    mean = mean / w
    cov = cov / w - np.outer(mean, mean)
    return G(mean, cov, w)


def phdnavigator():
    I3 = np.eye(3)
    out = {"source": "mono-rfs-lib/Test/PHDNavigatorTest.cs", "tolerance": 1e-5, "params": LINEAR2D,
           "pose": [1.0, 2.0]}

    out["predict_initial"] = {   # :85-104
        "measurements": [[2, 3]], "model": [],
        "expected": [{"w": 0.05, "mean": [3, 5, 0], "cov": LINEAR2D["birth_covariance"]}]}
    out["predict_known"] = {     # :106-126
        "measurements": [[2, 3]], "model": [{"w": 1.0, "mean": [3, 5, 0], "cov": I3.tolist()}],
        "expected": [{"w": 1.0, "mean": [3, 5, 0], "cov": I3.tolist()}]}

    # Correct :128-193
    pd, clutter = 0.9, 3e-7
    comp1, comp2 = G([3, 5, 0], I3, 0.8), G([7, 5, 0], 4.0 * I3, 1.4)
    mcov = np.zeros((3, 3))
    mcov[:2, :2] = LINEAR2D["R"]
    gz1, gz2 = G([1 + 2, 2 + 3, 0], mcov, 1.0), G([1 + 5, 2 + 3, 0], mcov, 1.0)
    z11, z12, z21, z22 = multiply(gz1, comp1), multiply(gz1, comp2), multiply(gz2, comp1), multiply(gz2, comp2)
    s1, s2 = z11.w + z12.w, z21.w + z22.w
    out["correct"] = {
        "note": "expects all four (z, component) pairs: the radius gate must admit every component",
        "measurements": [[2, 3], [5, 3]],
        "model": [comp1.tojson(), comp2.tojson()],
        "expected": [comp1.tojson(0.8 * (1 - pd)), comp2.tojson(1.4 * (1 - pd)),
                     z11.tojson(z11.w * pd / (clutter + pd * s1)), z12.tojson(z12.w * pd / (clutter + pd * s1)),
                     z21.tojson(z21.w * pd / (clutter + pd * s2)), z22.tojson(z22.w * pd / (clutter + pd * s2))]}

    # Prune :195-265
    mw, md = LINEAR2D["min_weight"], LINEAR2D["merge_threshold"]
    big = [G([-12, -24, -54], I3, 23.0), G([-80, -22, -12], 4.0 * I3, 1.0), G([-63, -11, -95], 0.1 * I3, 6.0)]
    irrelevant = [G([12, 24, 54], I3, 0.3 * mw), G([80, 22, 12], 4.0 * I3, 0.8 * mw),
                  G([63, 11, 95], 0.1 * I3, 0.99 * mw), G([23, 19, 73], I3, 0.0 * mw)]
    m1 = [G([0, 0, 0], I3, 1.0), G([0, md, 0], I3, 0.6), G([0, md / 2, 0], I3, 1.2)]
    m2 = [G([99 - md / 6, 99, 99], I3, 0.9), G([99, 99 - md / 6, 99], I3, 0.5), G([99, 99, 99 - md / 6], I3, 1.1)]
    out["prune"] = {
        "model": [g.tojson() for g in big + irrelevant + m1 + m2],
        "expected": [g.tojson() for g in big] + [merge(m1).tojson(), merge(m2).tojson()]}
    return out


def dense(n, entries, default=0.0):
    m = [[default] * n for _ in range(n)]
    for (i, k), v in entries.items():
        m[i][k] = v
    return m


def graphcombinatorics():
    three = {(0, 0): 1, (1, 0): 1, (1, 1): 1, (2, 2): 1, (2, 3): 1, (2, 4): 1, (3, 3): 1, (4, 4): 1, (5, 5): 1}  # :49-64
    m3 = [[6, 8, 5], [7, 3, 4], [9, 8, 7]]
    out = {"source": "mono-rfs-lib/Test/GraphCombinatoricsTest.cs",
           "note": "dense matrices; `defined` lists the explicitly set entries, other entries hold the "
                   "SparseMatrix default of the test (0)"}

    def cc(entries, n=6):
        return {"n": n, "defined": sorted([list(k) for k in entries])}

    two = dict(three); two[(1, 2)] = 1
    one = dict(two); one[(5, 4)] = 1
    out["connected_components"] = [
        {"name": "empty", "n": 100, "defined": [], "count": 0},                                   # :66-76
        {"name": "full10", "n": 10, "defined": [[i, k] for i in range(10) for k in range(10)], "count": 1},  # :78-95
        dict(cc(three), name="three", count=3), dict(cc(two), name="two", count=2), dict(cc(one), name="one", count=1)]  # :97-129

    av2 = dict(three); av2[(1, 0)] = 100
    out["assignment_value"] = [
        {"matrix": dense(6, three), "matches": [0, 1, 2, 3, 4, 5], "expected": 6},                # :174-181
        {"matrix": dense(6, av2), "matches": [1, 0, 4, 0, 4, 5], "expected": 103}]                # :183-198

    la3 = dict(three); la3[(4, 2)] = 3; del la3[(2, 2)]
    out["linear_assignment"] = [
        {"name": "unique", "matrix": dense(10, {(i, i): (i + 1) / 2.0 for i in range(10)}), "expected": list(range(10))},  # :200-214
        {"name": "1", "matrix": m3, "expected": [1, 0, 2]},                                       # :216-229
        {"name": "2", "matrix": dense(3, {(0, 1): 2, (0, 2): 5, (1, 0): 3, (1, 2): 6, (2, 0): 1, (2, 1): 2}), "expected": [2, 0, 1]},  # :231-244
        {"name": "3", "matrix": dense(6, la3), "expected": [0, 1, 4, 3, 2, 5]}]                   # :246-255

    out["lexicographical"] = [
        {"matrix": m3, "modelsize": 3, "expected": [[0, 1, 2], [0, 2, 1], [1, 0, 2], [1, 2, 0], [2, 0, 1], [2, 1, 0]]},   # :257-281
        {"matrix": m3, "modelsize": 1, "expected": [[0, 2, 1], [1, 2, 0], [2, 1, 0]]}]            # :283-306

    out["murty_children"] = [
        {"forced": [[1, 1]], "eliminated": [[0, 2]], "assignment": [0, 1, 2, 3, 4],               # :308-334
         "expected": [{"forced": [[1, 1]], "eliminated": [[0, 2], [0, 0]]},
                      {"forced": [[1, 1], [0, 0]], "eliminated": [[0, 2], [2, 2]]},
                      {"forced": [[1, 1], [0, 0], [2, 2]], "eliminated": [[0, 2], [3, 3]]}]},
        {"forced": [[0, 0], [1, 1], [2, 2], [3, 3], [4, 4]], "eliminated": [[1, 2]], "assignment": [0, 1, 2, 3, 4],  # :336-355
         "expected": []}]

    ninf = "-inf"
    out["murty_pairing"] = [
        {"matrix": m3, "expected": [[1, 0, 2], [1, 2, 0], [2, 0, 1], [0, 2, 1], [2, 1, 0], [0, 1, 2]]},  # :357-382
        # :384-404; reduceprofit rebuilds with default -inf (GraphCombinatorics.cs:208) so only the diagonal exists
        {"matrix": [[1 if i == k else ninf for k in range(5)] for i in range(5)], "expected": [[0, 1, 2, 3, 4]]}]
    return out


def resample():
    return {"source": "mono-rfs-lib/Test/SimulationTest.cs:225-270",
            "weights": [0.11, 0.28, 0.31, 0.01, 0.29],
            "best_source": 2, "always_present": [1, 2, 4], "sometimes_absent": [0, 3]}


if __name__ == "__main__":
    for name, data in (("phdnavigator_kat", phdnavigator()), ("graphcombinatorics_kat", graphcombinatorics()),
                       ("resample_kat", resample())):
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(data, f, indent=1)
        print("wrote", name)
