#!/usr/bin/env python3
"""Registers, scratch, occupancy and LDS of every kernel, from hipcc's -Rpass-analysis=kernel-resource-usage remarks:
    hipcc ... -Rpass-analysis=kernel-resource-usage -o x.so phdhip.hip 2> build.log; python scripts/kernel_resources.py build.log"""
import re
import subprocess
import sys


def main(path):
    txt = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)
    rows = []
    for b in blocks[1:]:
        name = b.split("\n")[0].strip().split()[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        dem = [r[0] for r in rows]
    print("%-72s %5s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
    for r, n in zip(rows, dem):
        n = re.sub(r"\(DevParams.*", "", n).replace("void ", "")
        print("%-72s %5d %5d %5d %7d %4d %7d" % ((n[:72],) + r[1:]))


if __name__ == "__main__":
    main(sys.argv[1])
