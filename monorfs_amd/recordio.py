"""The reference's record / replay wire formats for the Python host mirror; the C++ host uses
monorfs_amd/host/RecordIO.hpp, which this file follows function by function.

Readers: mono-rfs-lib/Util/FileParser.cs, Util.ParseDictionary (Util.cs:232-264), SimulatedVehicle.FromFile
(SimulatedVehicle.cs:346-385). Writers: Simulation.Serialized* (Simulation.cs:155-231), Gaussian.ToString("g6")
(Gaussian.cs:391-431). A record is the zip written by Simulation.SaveToFile (Simulation.cs:391-488) or the directory of
its members: scene.world trajectory.out odometry.out measurements.out estimate.out maps.out tags.out config.cfg.
Also the command file of `-c=` (FileParser.CommandsFromDescriptor) and the configuration file of `-g=` (Config.cs)."""
import os
import zipfile

import numpy as np


class FormatError(ValueError):
    pass


def _double(token, error):
    try:
        return float(token)
    except ValueError:
        raise FormatError(error)


def parse_double_list(descriptor):                       # FileParser.cs:279-294
    return [_double(v, "the double descriptor '%s' is malformed" % descriptor) for v in descriptor.split(" ") if v != ""]


def timed_array_from_descriptor(lines, dim):             # :104-119
    out = []
    for line in lines:
        v = parse_double_list(line)
        if len(v) != dim + 1:
            raise FormatError("wrong state dimension")
        out.append((v[0], np.array(v[1:])))
    return out


def measurements_from_descriptor(descriptor, dim):       # :179-230
    history = []
    for frame in descriptor.split("\n"):
        parts = frame.split(":")
        if len(parts) != 2:
            raise FormatError("bad measurement format: no ':' delimiter found")
        time = _double(parts[0], "bad measurement format: missing time")
        points = []
        for point in parts[1].split(";"):
            if point == "":
                continue
            comps = point.split(" ")
            if len(comps) != dim:
                raise FormatError("wrong measurement dimension")
            points.append([_double(c, "bad measurement format: invalid point") for c in comps])
        history.append((time, np.array(points, float).reshape(-1, dim)))
    return history


def parse_gaussian_descriptor(descriptor, dim=3):        # :302-339
    bad = "the double descriptor '%s' is malformed" % descriptor
    parts = descriptor.split(";")
    if len(parts) < 3:
        raise FormatError(bad)
    w = _double(parts[0], bad)
    mean, cov = parts[1].split(" "), parts[2].split(" ")
    if len(cov) != len(mean) ** 2:
        raise FormatError("covariance has the wrong size")
    if len(mean) != dim:
        raise FormatError("wrong gaussian dimension")
    return w, np.array([_double(v, bad) for v in mean]), np.array([_double(v, bad) for v in cov]).reshape(dim, dim)


def map_history_from_descriptor(descriptor, dim=3):      # :128-170
    history = []
    for frame in (f for f in descriptor.split("\n|\n") if f != ""):
        lines = [l for l in frame.split("\n") if l != ""]
        time = _double(lines[0], "bad map format: missing time") if lines else None
        if time is None:
            raise FormatError("bad map format: missing time")
        comps = [parse_gaussian_descriptor(l, dim) for l in lines[1:]]
        history.append((time, (np.array([c[0] for c in comps]), np.array([c[1] for c in comps]).reshape(-1, dim),
                               np.array([c[2] for c in comps]).reshape(-1, dim, dim))))
    return history


def trajectory_history_from_descriptor(descriptor, dim, filterhistory=False):   # :65-95
    history, filtered = [], []
    for frame in (f for f in descriptor.split("\n|\n") if f != ""):
        lines = [l for l in frame.split("\n") if l != ""]
        if not lines:
            raise FormatError("bad trajectory format: missing time")
        time = _double(lines[0], "bad trajectory format: missing time")
        trajectory = timed_array_from_descriptor(lines[1:], dim)
        if filterhistory:
            filtered.append(trajectory[-1])
            history.append((time, list(filtered)))
        else:
            history.append((time, trajectory))
    return history


def parse_dictionary(descriptor):                        # Util.cs:232-264
    lines = descriptor.replace("\r\n", "\n").replace("\r", "\n").split("\n")
    out, key = {}, ""
    if lines and (lines[0] == "" or lines[0][0] in " \t"):
        return out
    for line in lines:
        if line.strip() == "":
            continue
        if line[0] != "\t":
            key = line
            out[key] = []
        else:
            out[key].append(line[1:])
    return out


def scene_from_descriptor(descriptor):                   # SimulatedVehicle.cs:346-385
    d = parse_dictionary(descriptor)
    pose = np.array(parse_double_list(d["pose"][0]))
    key = "focal" if "focal" in d else ("params" if "params" in d else "")
    params = np.array(parse_double_list(d[key][0])) if key else None
    landmarks = []
    for line in d["landmarks"]:
        lm = parse_double_list(line)
        if len(lm) != 3:
            raise FormatError("Map landmarks must be 3D")
        landmarks.append(lm)
    return pose, params, np.array(landmarks, float).reshape(-1, 3)


def g6(x):                                               # double.ToString("g6")
    return "%.6g" % x


def g15(x):                                              # double.ToString()
    return "%.15g" % x


def gaussian_to_string(w, mean, cov):                    # Gaussian.cs:391-431
    return g6(w) + ";" + " ".join(g6(v) for v in mean) + ";" + " ".join(g6(v) for v in np.asarray(cov).reshape(-1))


def serialize_scene(pose, params, landmarks):            # Vehicle.ToString("g6"), Vehicle.cs:513-524 (scene.world, -f=)
    return ("pose\n\t" + " ".join(g6(v) for v in pose) + "\nparams\n\t" + " ".join(g6(v) for v in params) +
            "\nlandmarks\n\t" + "\n\t".join(" ".join(g6(v) for v in l) for l in landmarks) + "\n")


def serialize_timed_array(a):                            # Simulation.cs:155-166, 225-231
    return "\n".join(g6(t) + "".join(" " + g6(v) for v in vec) for t, vec in a)


def serialize_measurements(m):                           # :186-193
    return "\n".join(g6(t) + ":" + ";".join(" ".join(g15(c) for c in p) for p in pts) for t, pts in m)


def serialize_maps(maps):                                # :199-206
    return "\n|\n".join(g6(t) + "".join("\n" + gaussian_to_string(w, m, c) for w, m, c in zip(*mix)) for t, mix in maps)


def serialize_trajectories(t):                           # :172-181
    return "\n|\n".join(g6(time) + "\n" + serialize_timed_array(traj) for time, traj in t)


def timed_message_from_descriptor(lines):                # FileParser.cs:237-256 (tags.out)
    out = []
    for line in lines:
        values = line.split(" ", 1)
        t = _double(values[0], "the TimedMessage descriptor '%s' is malformed" % line)
        if len(values) < 2:
            raise IndexError("the TimedMessage descriptor '%s' has no message" % line)   # values[1] in the reference
        out.append((t, values[1]))
    return out


def serialize_tags(tags):                                # Manipulator.SerializedTags, Manipulator.cs:294-304
    return "\n".join(g6(t) + " " + message for t, message in tags)


def commands_from_descriptor(lines):                     # FileParser.cs:263-274 (-c=moves.in)
    """one line per frame: odometry (6 values for Pose3D) [+ 7th > 0: start SLAM, < 0: start mapping] [+ screenshot flag
    and camera theta phi zoom] (Simulation.cs:575-607)"""
    return [parse_double_list(line) for line in lines]


# ---- Config (mono-rfs-lib/Config.cs): `FieldName: value` lines, matrices in Octave syntax --------------------------------
# field -> type, in declaration order (Config.cs:46-104): the order Config.ToString() writes them in
CONFIG_FIELDS = (
    ("NParallel", "int"), ("Model", "enum"), ("AxisLimit", "double"), ("MeasureElapsed", "timespan"), ("MapClip", "vector"),
    ("UseOdometry", "bool"), ("CheckpointCycleTime", "int"), ("MotionCovariance", "matrix"), ("MeasurementCovariance", "matrix"),
    ("DetectionProbability", "double"), ("ClutterDensity", "double"), ("PerfectStill", "bool"), ("VisibilityRamp", "vector"),
    ("KinectDelta", "int"), ("KeypointFilter", "bool"), ("ShowVisible", "bool"), ("DensityDistanceThreshold", "double"),
    ("BirthCovariance", "matrix"), ("BirthWeight", "double"), ("MinWeight", "double"), ("MinEffectiveParticle", "double"),
    ("MaxQuantity", "int"), ("MergeThreshold", "double"), ("ExplorationThreshold", "double"), ("RenderAllParticles", "bool"),
    ("MotionCovarianceMultiplier", "double"), ("MeasurementCovarianceMultiplier", "double"), ("NavigatorPD", "double"),
    ("NavigatorClutterDensity", "double"), ("GradientAscentRate", "double"), ("GradientClip", "double"),
    ("MatchThreshold", "double"), ("NewLandmarkThreshold", "int"), ("DAAlgorithm", "enum"), ("OdometryMergeThreshold", "double"),
)


def default_config():
    """the static initialisers of Config followed by SetPRM3DDefaults (Config.cs:46-111, 238-263)"""
    R = [[2.0, 0, 0], [0, 2.0, 0], [0, 0, 1e-3]]
    return {
        "NParallel": 8, "Model": "PRM3D", "AxisLimit": 10.0, "MeasureElapsed": 0.0333333,   # new TimeSpan(10000000 / 30) ticks
         "MapClip": [-6.0, 6.0, -3.0, 3.0],
        "UseOdometry": True, "CheckpointCycleTime": 300,
        "MotionCovariance": [[5e-3 if i == k and i < 3 else (2e-4 if i == k else 0.0) for k in range(6)] for i in range(6)],
        "MeasurementCovariance": R, "DetectionProbability": 0.9, "ClutterDensity": 3e-7, "PerfectStill": False,
        "VisibilityRamp": [3 * R[0][0] ** 0.5, 3 * R[1][1] ** 0.5, 3 * R[2][2] ** 0.5],
        "KinectDelta": 4, "KeypointFilter": True, "ShowVisible": False, "DensityDistanceThreshold": 0.5,
        "BirthCovariance": [[1e-2, 0, 0], [0, 1e-2, 0], [0, 0, 1e-2]], "BirthWeight": 0.05, "MinWeight": 1e-3,
        "MinEffectiveParticle": 0.1, "MaxQuantity": 600, "MergeThreshold": 0.3, "ExplorationThreshold": 1e-5,
        "RenderAllParticles": True, "MotionCovarianceMultiplier": 1.0, "MeasurementCovarianceMultiplier": 1.0,
        "NavigatorPD": 0.9, "NavigatorClutterDensity": 3e-7, "GradientAscentRate": 1e-2, "GradientClip": 10.0,
        "MatchThreshold": 3.0, "NewLandmarkThreshold": 3, "DAAlgorithm": "Mahalanobis", "OdometryMergeThreshold": 1e-2,
    }


def parse_octave_matrix(text):
    """`[a b; c d]` (Accord's OctaveMatrixFormatProvider, the syntax Config.FromDescriptor hands to Matrix.ParseJagged)"""
    t = text.strip()
    if t.startswith("["):
        t = t[1:]
    if t.endswith("]"):
        t = t[:-1]
    rows = [r for r in (row.strip() for row in t.split(";")) if r != ""]
    return [[_double(v, "the matrix descriptor '%s' is malformed" % text) for v in row.replace(",", " ").split()] for row in rows]


def config_from_descriptor(lines, config=None, log=None):
    """Config.FromDescriptor (Config.cs:155-209): one `FieldName: value` per line, unknown fields ignored, any missing
    parameter left as it is (`config`, default: default_config()); a line without a colon is reported and skipped."""
    cfg = dict(default_config() if config is None else config)
    types = dict(CONFIG_FIELDS)
    for line in lines:
        pair = line.split(":", 1)
        if len(pair) != 2:
            if log:
                log("Skipped malformed configuration line:\n'" + line + "'")
            continue
        name, value = pair[0].strip(), pair[1].strip()
        kind = types.get(name)
        if kind is None:
            continue
        if kind == "matrix":
            cfg[name] = parse_octave_matrix(value)
        elif kind == "vector":
            cfg[name] = parse_octave_matrix(value)[0]
        elif kind == "timespan":   # new TimeSpan((long) (10000000 * seconds)): whole ticks of 100 ns
            cfg[name] = int(1e7 * _double(value, "Input string was not in a correct format.")) / 1e7
        elif kind == "double":
            cfg[name] = _double(value, "Input string was not in a correct format.")
        elif kind == "int":
            try:
                cfg[name] = int(value)
            except ValueError:
                raise FormatError("Input string was not in a correct format.")
        elif kind == "bool":
            if value.lower() not in ("true", "false"):
                raise FormatError("String was not recognized as a valid Boolean.")
            cfg[name] = value.lower() == "true"
        else:
            cfg[name] = value
    return cfg


def _octave(value):
    if len(value) and isinstance(value[0], (list, tuple, np.ndarray)):
        return "[" + "; ".join(" ".join(g15(v) for v in row) for row in value) + "]"
    return "[" + " ".join(g15(v) for v in value) + "]"


def serialize_config(config):
    """Config.ToString (Config.cs:268-309): every field, `Name: value`"""
    lines = []
    for name, kind in CONFIG_FIELDS:
        v = config[name]
        if kind in ("matrix", "vector"):
            r = _octave(v)
        elif kind == "bool":
            r = "True" if v else "False"
        elif kind in ("double", "timespan"):
            r = g15(v).replace("e-0", "E-0").replace("e+", "E+").replace("e-", "E-")
        else:
            r = str(v)
        lines.append(name + ": " + r)
    return "\n".join(lines)


def phd_params_from_config(config, measurer=None, **caps):
    """The values PHDNavigator reads from Config, as the parameter block of libphdhip (include/phdhip.h): its particles
    are clones of the reference vehicle with MeasurementCovarianceMultiplier, NavigatorPD and NavigatorClutterDensity
    (PHDNavigator.cs:257-259). `measurer`: the 7 values of PRM3DMeasurer.ToLinear() (default: `new PRM3DMeasurer()`)."""
    from .abi import prm3d_defaults
    if config.get("Model", "PRM3D") != "PRM3D":
        raise FormatError("libphdhip implements the PRM3D model")
    p = prm3d_defaults(**caps)
    if measurer is not None:
        p.measurer[:] = list(measurer)
    k = config["MeasurementCovarianceMultiplier"]
    p.R[:] = [k * v for row in config["MeasurementCovariance"] for v in row]
    p.visibility_ramp[:] = list(config["VisibilityRamp"])[:3]
    p.pd, p.clutter_density = config["NavigatorPD"], config["NavigatorClutterDensity"]
    p.birth_covariance[:] = [v for row in config["BirthCovariance"] for v in row]
    p.birth_weight, p.min_weight = config["BirthWeight"], config["MinWeight"]
    p.min_effective_particle, p.max_quantity = config["MinEffectiveParticle"], int(config["MaxQuantity"])
    p.merge_threshold, p.exploration_threshold = config["MergeThreshold"], config["ExplorationThreshold"]
    p.density_distance_threshold = config["DensityDistanceThreshold"]
    if "max_components" not in caps:
        p.max_components = max(p.max_components, p.max_quantity)
    return p


MEMBERS = ("scene.world", "trajectory.out", "odometry.out", "measurements.out", "estimate.out", "maps.out", "tags.out", "config.cfg")


def read_record(path):
    """member name -> text, from the zip of Simulation.SaveToFile or a directory holding its members"""
    out = {}
    if os.path.isdir(path):
        for name in MEMBERS:
            f = os.path.join(path, name)
            if os.path.exists(f):
                with open(f) as fh:
                    out[name] = fh.read()
    else:
        with zipfile.ZipFile(path) as z:
            for name in z.namelist():
                if name in MEMBERS:
                    out[name] = z.read(name).decode()
    return out


def write_record(path, members):
    if path.endswith(".zip"):
        with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
            for name, text in members.items():
                z.writestr(name, text)
    else:
        os.makedirs(path, exist_ok=True)
        for name, text in members.items():
            with open(os.path.join(path, name), "w") as fh:
                fh.write(text)
