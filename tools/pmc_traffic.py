"""Per-launch HBM traffic of one kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write rollout_stream_kernel profiles/round1_rollout_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so it is doubled (checked against this repo's own kernels: the Stein pass reads theta + score once per
XCD = 1.97 MB and reports 1.06 MB); WRITE_SIZE is exact (noise_fill_kernel writes 125 829 120 B and reports 122 880 KiB).
"""
import csv
import glob
import json
import sys


def mean_counter(d, name, kernel_substr):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and kernel_substr in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


def main():
    dfetch, dwrite, kernel, out = sys.argv[1:5]
    family = sys.argv[5] if len(sys.argv) > 5 else None  # tools/srcstamp.py family: the summary is stamped with the sources it was measured on
    fk, nf = mean_counter(dfetch, "FETCH_SIZE", kernel)
    wk, nw = mean_counter(dwrite, "WRITE_SIZE", kernel)
    res = dict(kernel=kernel, launches_fetch_pass=nf, launches_write_pass=nw, FETCH_SIZE_KiB_raw=fk, WRITE_SIZE_KiB_raw=wk,
               fetch_bytes_corrected=2.0 * fk * 1024, write_bytes=wk * 1024, hbm_bytes_per_launch=2.0 * fk * 1024 + wk * 1024,
               corrections="FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B); WRITE_SIZE x1; KiB -> bytes")
    if family:
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import srcstamp

        res["source_family"], res["source_stamp"] = family, srcstamp.stamp(family)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
