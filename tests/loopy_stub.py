"""An oracle-backed stand-in for PHDNavigator's batch evaluations, so that the host logic of monorfs_amd/loopy.py can
be tested without a GPU and the device can be compared with the oracle through whole pose searches."""
import numpy as np

import orc


class OracleNav:
    def __init__(self, params):
        self.params = params
        self.calls = 0

    def QuasiSetLogLikelihood(self, measurements, landmarks, poses):
        poses = np.asarray(poses, float).reshape(-1, 7)
        assert len(poses) <= self.params.max_particles
        self.calls += 1
        return np.array([orc.quasi_set_log_likelihood(self.params, q, landmarks, measurements) for q in poses])

    def QuasiSetLogLikelihoodGradient(self, measurements, landmarks, poses, average_mode=0):
        poses = np.asarray(poses, float).reshape(-1, 7)
        assert len(poses) <= self.params.max_particles
        self.calls += 1
        r = [orc.quasi_set_log_likelihood_grad(self.params, q, landmarks, measurements, average_mode) for q in poses]
        return np.array([v for v, _ in r]), np.array([g for _, g in r]).reshape(-1, 6)


def scene(rng, params, J, M, sigma=1.0):
    """a map of J unit-weight landmarks in view of the identity pose and M measurements of the first ones"""
    from test_oracle_crosscheck import measure_to_map
    pose = np.array([0, 0, 0, 1.0, 0, 0, 0])
    zs = np.column_stack([rng.uniform(-250, 250, J), rng.uniform(-180, 180, J), rng.uniform(0.5, 1.6, J)])
    lm = np.array([measure_to_map(params, pose, zz) for zz in zs])
    R = np.sqrt(np.diag(np.array(params.R).reshape(3, 3)))
    z = zs[:M] + rng.normal(0, sigma, (M, 3)) * R
    return pose, lm, z
