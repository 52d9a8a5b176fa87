"""Pose3D arithmetic of the reference for the Python host tools (BaseStructures/Poses/Pose3D.cs, Quaternion.cs): a pose
is x y z qw qx qy qz, an odometry / linear delta dx dy dz and a rotation vector."""
import math

import numpy as np


def qmul(a, b):   # Quaternion.cs:295-301
    return np.array([a[0] * b[0] - (a[1] * b[1] + a[2] * b[2] + a[3] * b[3]),
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                     a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qexp(lie):   # Quaternion.Exp, Quaternion.cs:185-196
    lie = np.asarray(lie, float)
    phi = np.linalg.norm(lie)
    if phi < 1e-12:
        return np.array([1.0, 0, 0, 0])
    return np.concatenate([[math.cos(phi)], math.sin(phi) * (lie / phi)])


def qlog(q):   # Quaternion.Log, Quaternion.cs:204-218
    q = np.asarray(q, float) / np.linalg.norm(q)
    phi = math.acos(min(1.0, max(-1.0, q[0])))
    mag = np.linalg.norm(q[1:])
    if mag < 1e-12:
        return np.zeros(3)
    return phi * (q[1:] / mag)


def qsqrt(q):   # Quaternion.Sqrt, Quaternion.cs:226-236
    if abs(q[0] + 1.0) < 1e-8:
        return np.array([1.0, 0, 0, 0])
    rw = math.sqrt(0.5 * (1 + q[0]))
    alpha = 1 / (2 * rw)
    return np.array([rw, alpha * q[1], alpha * q[2], alpha * q[3]])


def _rot(q, v):
    return qmul(qmul(q, np.concatenate([[0.0], np.asarray(v, float)])), qconj(q))[1:]


def add_odometry(pose7, delta6):
    """Pose3D.AddOdometry (Pose3D.cs:314-333): the translation applied in the frame halfway through the rotation"""
    pose7, delta6 = np.asarray(pose7, float), np.asarray(delta6, float)
    q = pose7[3:]
    dori = qexp(0.5 * delta6[3:])                     # Identity.FromLinear
    mid = qmul(q, qsqrt(dori))
    nq = qmul(q, dori)
    return np.concatenate([pose7[:3] + _rot(mid, delta6[:3]), nq / np.linalg.norm(nq)])


def diff_odometry(pose7, origin7):
    """Pose3D.DiffOdometry (Pose3D.cs:338-356): the odometry that takes `origin` into `pose`"""
    pose7, origin7 = np.asarray(pose7, float), np.asarray(origin7, float)
    qo = origin7[3:]
    dq = qmul(qconj(qo), pose7[3:])
    mid = qmul(qo, qsqrt(dq))
    dx = _rot(qconj(mid), pose7[:3] - origin7[:3])
    return np.concatenate([dx, 2 * qlog(dq)])          # dq.ToLinear()
