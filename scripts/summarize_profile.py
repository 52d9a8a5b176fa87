#!/usr/bin/env python3
"""Summarise a gpurun_out/prof_<tag>/ directory written by scripts/profile_gpu.sh into
profiles/<name>.md + profiles/<name>.json (per-kernel mean duration from the kernel trace, PMC
counters per launch, HBM traffic with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    import re
    name = name.replace("void ", "").split("(")[0]
    return re.sub(r"<.*>$", "", name)


def newest(pattern):
    """gpurun merges a run's files into what earlier runs left under the same directory (the file names carry the process
    id): only the newest file of a kind is this run's."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


def load_counters(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in newest(os.path.join(d, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            out[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def main():
    src, name = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stats = {}
    for f in newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            stats[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                                         "min_us": float(row["MinNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3,
                                         "pct": float(row["Percentage"])}
    res = {}
    for f in newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv")):
        for row in csv.DictReader(open(f)):
            res.setdefault(short(row["Kernel_Name"]), {"vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]),
                                                        "sgpr": int(row["SGPR_Count"]), "lds": int(row["LDS_Block_Size"]),
                                                        "wg": int(row["Workgroup_Size_X"]), "grid": int(row["Grid_Size_X"])})
    pmc = {}
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        for k, cs in load_counters(os.path.join(src, sub)).items():
            for c, vals in cs.items():
                pmc.setdefault(k, {})[c] = sum(vals) / len(vals)
    out = {"source": src, "kernels": {}}
    lines = ["# %s" % name, "", "rocprofv3 --kernel-trace --stats + separate --pmc passes (scripts/profile_gpu.sh); values are per launch.",
             "HBM bytes = 2 x FETCH_SIZE x 1024 (gfx950 reports half of a wide streaming read) + WRITE_SIZE x 1024.", "",
             "| kernel | calls | avg us | % | VGPR | LDS B | grid x wg | HBM read MB | HBM write MB | VALU insts/wave | VALU busy frac | wait-any frac | L2 hit |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, st in sorted(stats.items(), key=lambda kv: -kv[1]["pct"]):
        if not k.startswith("k_"):
            continue
        c = pmc.get(k, {})
        r = res.get(k, {})
        rd = 2 * c.get("FETCH_SIZE", 0) * 1024 / 1e6 if "FETCH_SIZE" in c else None
        wr = c.get("WRITE_SIZE", 0) * 1024 / 1e6 if "WRITE_SIZE" in c else None
        waves = c.get("SQ_WAVES", 0)
        wc = c.get("SQ_WAVE_CYCLES", 0)
        valu_per_wave = c.get("SQ_INSTS_VALU", 0) / waves if waves else None
        busy = c.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else None
        wait = c.get("SQ_WAIT_ANY", 0) / wc if wc else None
        hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1) if "TCC_HIT_sum" in c else None
        out["kernels"][k] = {"trace": st, "resources": r, "pmc": c, "hbm_read_bytes": None if rd is None else rd * 1e6,
                             "hbm_write_bytes": None if wr is None else wr * 1e6}
        f = lambda v, fmt="%.2f": "-" if v is None else fmt % v
        lines.append("| %s | %d | %.1f | %.1f | %s | %s | %sx%s | %s | %s | %s | %s | %s | %s |" % (
            k, st["calls"], st["avg_us"], st["pct"], r.get("vgpr", "-"), r.get("lds", "-"), r.get("grid", "-"), r.get("wg", "-"),
            f(rd), f(wr), f(valu_per_wave, "%.0f"), f(busy), f(wait), f(hit)))
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    open(os.path.join(root, "profiles", name + ".md"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(root, "profiles", name + ".json"), "w"), indent=1)
    # per-launch HBM bytes of every kernel, read by bench.py for `roofline.traffic`
    if len(sys.argv) > 3:
        tfile = os.path.join(root, "profiles", name.split("_")[0] + "_hbm_traffic.json")   # r02_...: from a one-stream profile, per whole-range launch
        traffic = json.load(open(tfile)) if os.path.exists(tfile) else {}
        traffic[sys.argv[3]] = {k: (v["hbm_read_bytes"] or 0) + (v["hbm_write_bytes"] or 0) for k, v in out["kernels"].items()
                                if v["hbm_read_bytes"] is not None}
        traffic["_note"] = "HBM bytes per launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (rocprofv3 --pmc, separate passes; gfx950 FETCH_SIZE correction)"
        json.dump(traffic, open(tfile, "w"), indent=1)
        # vector-ALU instructions per launch (SQ_INSTS_VALU summed over the waves), read by bench.py for the issue-rate view
        vfile = os.path.join(root, "profiles", name.split("_")[0] + "_valu_insts.json")
        valu = json.load(open(vfile)) if os.path.exists(vfile) else {}
        valu[sys.argv[3]] = {}
        for k, v in out["kernels"].items():
            n = v.get("pmc", {}).get("SQ_INSTS_VALU")
            wgs = (v.get("resources", {}).get("grid") or 0) // max(v.get("resources", {}).get("wg") or 1, 1)
            if n:   # one workgroup per particle in every kernel but the single-workgroup normalise
                valu[sys.argv[3]][k] = {"per_particle": n / wgs} if wgs >= 64 else {"per_launch": n}
        valu["_note"] = "wave-level vector-ALU instructions (rocprofv3 --pmc SQ_INSTS_VALU) per particle (= workgroup) or per launch"
        json.dump(valu, open(vfile, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
