"""Mean of every collected counter per kernel from one or more rocprofv3 --pmc passes.

  python tools/pmc_summary.py out.json dir_pass1 [dir_pass2 ...]

FETCH_SIZE / WRITE_SIZE are reported raw (KiB; FETCH_SIZE needs the gfx950 x2 correction, see tools/pmc_traffic.py);
SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = MFMA flops of the launch.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in acc.items():
        e = {}
        for name, v in sorted(cs.items()):
            e[name + "_mean"] = sum(v) / len(v)
            e[name + "_launches"] = len(v)
        if e.get("SQ_INSTS_VALU_MFMA_MOPS_F32_mean"):
            e["mfma_flops_per_launch"] = 512.0 * e["SQ_INSTS_VALU_MFMA_MOPS_F32_mean"]
        res[k] = e
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res.items():
        print(k[:90], {n: round(v, 1) for n, v in e.items() if n.endswith("_mean")})


if __name__ == "__main__":
    main()
