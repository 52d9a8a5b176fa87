"""ctypes view of include/phdhip.h: the parameter block and constants shared by every binding.

Field order and types must match `struct phd_params` exactly (include/phdhip.h)."""
import ctypes as C
import math

PHD_API_VERSION = 4

PHD_OK, PHD_ERR_GENERIC = 0, -1
PHD_ERR_BAD_ARGUMENT, PHD_ERR_CAPACITY, PHD_ERR_ASSOCIATION, PHD_ERR_DEVICE, PHD_ERR_NO_DEVICE = 1, 2, 3, 4, 5

PHD_MODEL_LINEAR2D, PHD_MODEL_PRM3D = 0, 1
PHD_GATE_EUCLIDEAN, PHD_GATE_SQUARED_EUCLIDEAN, PHD_GATE_DISABLED = 0, 1, 2

PHD_STAGE_PREDICTED, PHD_STAGE_CORRECTED, PHD_STAGE_PRUNED = 0, 1, 2


class PhdParams(C.Structure):
    _fields_ = [
        ("model", C.c_int32),
        ("zdim", C.c_int32),
        ("measurer", C.c_double * 7),
        ("R", C.c_double * 9),
        ("visibility_ramp", C.c_double * 3),
        ("pd", C.c_double),
        ("clutter_density", C.c_double),
        ("birth_covariance", C.c_double * 9),
        ("birth_weight", C.c_double),
        ("min_weight", C.c_double),
        ("min_effective_particle", C.c_double),
        ("max_quantity", C.c_int32),
        ("gate_metric", C.c_int32),
        ("merge_threshold", C.c_double),
        ("exploration_threshold", C.c_double),
        ("density_distance_threshold", C.c_double),
        ("max_particles", C.c_int32),
        ("max_components", C.c_int32),
        ("max_measurements", C.c_int32),
        ("emit_capacity", C.c_int32),
    ]


def prm3d_defaults(max_particles=1, max_components=600, max_measurements=64):
    """Config.SetPRM3DDefaults (Config.cs:238-263) + the PHD constants of Config.cs:74-91 +
    `new PRM3DMeasurer()` (PRM3DMeasurer.cs:70-73). Python twin of phd_default_params()."""
    import numpy as np
    p = PhdParams()
    p.model, p.zdim = PHD_MODEL_PRM3D, 3
    p.measurer[:] = [575.8156, float(np.float32(0.1)), 2.0, -320, -240, 640, 480]
    R = [2.0, 0, 0, 0, 2.0, 0, 0, 0, 1e-3]
    p.R[:] = R
    p.visibility_ramp[:] = [3 * math.sqrt(R[0]), 3 * math.sqrt(R[4]), 3 * math.sqrt(R[8])]
    p.pd, p.clutter_density = 0.9, 3e-7
    p.birth_covariance[:] = [1e-2, 0, 0, 0, 1e-2, 0, 0, 0, 1e-2]
    p.birth_weight, p.min_weight, p.min_effective_particle = 0.05, 1e-3, 0.1
    p.max_quantity, p.gate_metric = 600, PHD_GATE_SQUARED_EUCLIDEAN
    p.merge_threshold, p.exploration_threshold, p.density_distance_threshold = 0.3, 1e-5, 0.5
    p.max_particles, p.max_components, p.max_measurements = max_particles, max_components, max_measurements
    p.emit_capacity = 0
    return p


def params_from_dict(d, **caps):
    """Build a parameter block from the `params` object of a golden fixture."""
    p = PhdParams()
    p.model = PHD_MODEL_LINEAR2D if d["model"] == "linear2d" else PHD_MODEL_PRM3D
    p.zdim = d["zdim"]
    m = list(d["measurer"]) + [0.0] * (7 - len(d["measurer"]))
    p.measurer[:] = m
    R = [0.0] * 9
    for i in range(p.zdim):
        for k in range(p.zdim):
            R[i * p.zdim + k] = d["R"][i][k]
    p.R[:] = R
    ramp = list(d["visibility_ramp"]) + [0.0] * (3 - len(d["visibility_ramp"]))
    p.visibility_ramp[:] = ramp
    p.pd, p.clutter_density = d["pd"], d["clutter_density"]
    p.birth_covariance[:] = [x for row in d["birth_covariance"] for x in row]
    p.birth_weight, p.min_weight = d["birth_weight"], d["min_weight"]
    p.min_effective_particle, p.max_quantity = d["min_effective_particle"], d["max_quantity"]
    p.gate_metric = d.get("gate_metric", PHD_GATE_SQUARED_EUCLIDEAN)
    p.merge_threshold, p.exploration_threshold = d["merge_threshold"], d["exploration_threshold"]
    p.density_distance_threshold = d["density_distance_threshold"]
    p.max_particles = caps.get("max_particles", 1)
    p.max_components = caps.get("max_components", 600)
    p.max_measurements = caps.get("max_measurements", 64)
    p.emit_capacity = caps.get("emit_capacity", 0)
    return p
