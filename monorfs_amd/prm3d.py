"""The pixel-range measurer of the reference for the Python host tools (PRM3DMeasurer.cs), in numpy: what the
simulation and record-writing scripts need to produce measurements. The solver's own copy lives on the device
(csrc/phd_device.h); the tests check both against the oracle."""
import numpy as np

from .pose3d import qconj, qmul


def measure_perfect(pose7, x, focal):
    """PRM3DMeasurer.MeasurePerfect (PRM3DMeasurer.cs:138-149): landmark x seen from pose -> (px, py, range)"""
    pose7, x = np.asarray(pose7, float), np.asarray(x, float)
    diff = x - pose7[:3]
    q = pose7[3:]
    l = qmul(qmul(qconj(q), np.concatenate([[0.0], diff])), q)[1:]
    return np.array([focal * l[0] / l[2], focal * l[1] / l[2], np.sign(l[2]) * np.linalg.norm(diff)])


def measure_to_map(pose7, z, focal):
    """PRM3DMeasurer.MeasureToMap (PRM3DMeasurer.cs:299-312): measurement -> world point"""
    pose7, z = np.asarray(pose7, float), np.asarray(z, float)
    alpha = z[2] / np.sqrt(focal * focal + z[0] * z[0] + z[1] * z[1])
    d = np.array([alpha * z[0], alpha * z[1], alpha * focal])
    q = pose7[3:]
    return pose7[:3] + qmul(qmul(q, np.concatenate([[0.0], d])), qconj(q))[1:]


def fuzzy_visible(z, measurer7, ramp):
    """PRM3DMeasurer.FuzzyVisibleM (PRM3DMeasurer.cs:277-291): 0 .. 1; film rectangle of ints, float32 range clip"""
    rmin, rmax = float(np.float32(measurer7[1])), float(np.float32(measurer7[2]))
    left, top, width, height = (int(v) for v in measurer7[3:7])
    d = min((z[0] - left) / ramp[0], (left + width - z[0]) / ramp[0], (z[1] - top) / ramp[1], (top + height - z[1]) / ramp[1],
            (z[2] - rmin) / ramp[2], (rmax - z[2]) / ramp[2])
    return max(0.0, min(1.0, d))
