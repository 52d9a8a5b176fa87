"""The reference's record / replay wire formats for the Python host mirror; the C++ host uses
monorfs_amd/host/RecordIO.hpp, which this file follows function by function.

Readers: mono-rfs-lib/Util/FileParser.cs, Util.ParseDictionary (Util.cs:232-264), SimulatedVehicle.FromFile
(SimulatedVehicle.cs:346-385). Writers: Simulation.Serialized* (Simulation.cs:155-231), Gaussian.ToString("g6")
(Gaussian.cs:391-431). A record is the zip written by Simulation.SaveToFile (Simulation.cs:391-488) or the directory of
its members: scene.world trajectory.out odometry.out measurements.out estimate.out maps.out tags.out."""
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


def serialize_timed_array(a):                            # Simulation.cs:155-166, 225-231
    return "\n".join(g6(t) + "".join(" " + g6(v) for v in vec) for t, vec in a)


def serialize_measurements(m):                           # :186-193
    return "\n".join(g6(t) + ":" + ";".join(" ".join(g15(c) for c in p) for p in pts) for t, pts in m)


def serialize_maps(maps):                                # :199-206
    return "\n|\n".join(g6(t) + "".join("\n" + gaussian_to_string(w, m, c) for w, m, c in zip(*mix)) for t, mix in maps)


def serialize_trajectories(t):                           # :172-181
    return "\n|\n".join(g6(time) + "\n" + serialize_timed_array(traj) for time, traj in t)


MEMBERS = ("scene.world", "trajectory.out", "odometry.out", "measurements.out", "estimate.out", "maps.out", "tags.out")


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
