#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks and, with --guard, fail when a
kernel spills registers.

Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> remarks.txt
       tools/kernel_resources.py remarks.txt [--guard] [--max-sgpr N]

The guard exists because register spills in the group-divergent kernels of k_verify.hip / k_coalesce.hip (SGPR spills
travel through VGPR lanes with v_writelane/v_readlane; VGPR spills go to scratch under the current
exec mask) are the first suspect of a wrong-result build seen in round 1.
"""
import re
import sys

FIELDS = [
    ("sgpr", r"TotalSGPRs"),
    ("vgpr", r"VGPRs"),
    ("agpr", r"AGPRs"),
    ("scratch", r"ScratchSize \[bytes/lane\]"),
    ("occ", r"Occupancy \[waves/SIMD\]"),
    ("sgpr_spill", r"SGPRs Spill"),
    ("vgpr_spill", r"VGPRs Spill"),
    ("lds", r"LDS Size \[bytes/block\]"),
]


def parse(text):
    out = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = blk.split("\n")[0].strip().split(" ")[0]
        row = {"name": name}
        for key, pat in FIELDS:
            m = re.search(pat + r": (\d+)", blk)
            row[key] = int(m.group(1)) if m else -1
        out.append(row)
    return out


def demangle(names):
    import subprocess

    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        d = p.stdout.strip().split("\n")
        if len(d) == len(names):
            return d
    except OSError:
        pass
    return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    guard = "--guard" in sys.argv
    rows = parse(open(args[0]).read())
    names = demangle([r["name"] for r in rows])
    bad = []
    for r, n in zip(rows, names):
        n = n.replace("void ", "").replace("mtsv::", "").replace("(anonymous namespace)::", "")
        n = re.sub(r"\((?!anonymous).*$", "", n)
        print(f"{n[:44]:44s} SGPR {r['sgpr']:4d} VGPR {r['vgpr']:4d} scratch {r['scratch']:4d} occ {r['occ']:2d} "
              f"sgpr_spill {r['sgpr_spill']:3d} vgpr_spill {r['vgpr_spill']:3d} LDS {r['lds']}")
        if r["sgpr_spill"] > 0 or r["vgpr_spill"] > 0:
            bad.append(n)
    if guard and bad:
        print("register spills in: " + ", ".join(bad), file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
