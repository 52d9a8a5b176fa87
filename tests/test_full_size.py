"""BASELINE.json's full sizes on the GPU, through the C-ABI: size-independent properties of the step plus an
oracle check on a sample of particles (the whole oracle step at these sizes takes minutes on the host).
  B: 2048 particles x 512 components x 64 measurements      S: 4096 x 1024 x 128, MaxQuantity 1024"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import CONFIGS, Frame


def setup(cfg, profile):
    from monorfs_amd import navigator
    P, C, M, seed = CONFIGS[cfg]
    f = Frame(P, C, M, seed, weight_profile=profile)
    maxq = max(600, C)
    p = prm3d_defaults(max_particles=P, max_components=maxq, max_measurements=M)
    p.max_quantity = maxq
    nav = navigator.PHDNavigator(p, particlecount=P)
    nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    return nav, p, f


@pytest.mark.parametrize("cfg,profile", [("B", "steady"), ("B", "survey"), ("S", "steady")])
def test_full_size_properties_and_sample(cfg, profile):
    nav, p, f = setup(cfg, profile)
    nav.run_stages(f.z, with_alpha=True)
    alpha = nav.WeightAlpha()
    assert alpha.shape == (f.P,) and np.all(np.isfinite(alpha)) and np.all(alpha >= 0)
    rng = np.random.default_rng(1)
    for i in rng.choice(f.P, 6, replace=False):
        cw, cm, cc = nav.CorrectConditional(i)
        pw, pm, pc = nav.PruneModel(i)
        # every emitted weight reaches MinWeight; the pruned map holds at most MaxQuantity components, in
        # non-increasing order of the candidates' weights; merging conserves the weight of what it keeps
        assert np.all(cw >= p.min_weight)
        assert len(pw) <= p.max_quantity
        kept = np.sort(cw)[::-1][:p.max_quantity]
        assert np.isclose(pw.sum(), kept.sum(), rtol=1e-12)
        # covariances stay symmetric positive definite
        assert np.all(np.linalg.eigvalsh(pc) > 0)
        # oracle on this particle
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        opr = orc.prune(p, orc.correct(p, f.poses[i], f.z, pred))
        assert len(opr[0]) == len(pw)
        assert np.allclose(pw, opr[0], rtol=1e-7) and np.allclose(pm, opr[1], rtol=1e-7, atol=1e-11)
        iu = np.triu_indices(3)
        assert np.allclose(pc[:, iu[0], iu[1]], opr[2][:, iu[0], iu[1]], rtol=1e-7, atol=1e-13)
        a, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, opr)
        assert np.isclose(alpha[i], a, rtol=1e-6, atol=0)
    nav.close()


def test_full_size_step_resamples_consistently():
    nav, p, f = setup("B", "steady")
    nav.SlamUpdate(None, f.z, u_resample=0.41)
    w = nav.VehicleWeights
    src, res = nav.resample_sources()
    assert res, "the steady frame is expected to deplete the particle set"
    assert np.all(np.diff(src) >= 0) and src.min() >= 0 and src.max() < f.P     # systematic resampling is monotone
    assert np.allclose(w, 1.0 / f.P)
    # a resampled slot holds a copy of its source: equal maps for equal sources
    dup = np.flatnonzero(np.diff(src) == 0)
    assert len(dup) > 0
    i = int(dup[0])
    a, b = nav.MapModel(i), nav.MapModel(i + 1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert np.array_equal(nav.poses()[i], nav.poses()[i + 1])
    # idempotence of the frozen step: the same input gives the same output, bit for bit
    nav2, _, _ = setup("B", "steady")
    nav2.SlamUpdate(None, f.z, u_resample=0.41)
    src2, _ = nav2.resample_sources()
    assert np.array_equal(src, src2)
    m1, m2 = nav.MapModel(7), nav2.MapModel(7)
    assert all(np.array_equal(x, y) for x, y in zip(m1, m2))
    nav.close()
    nav2.close()
