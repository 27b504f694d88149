"""Per kernel: average duration (rocprofv3 --kernel-trace --stats CSV of one run) x VALU wave-instructions per launch (--pmc SQ_INSTS_VALU,
a separate run of the same command, summarised by tools/pmc_summary.py) -> the kernel's VALU issue fraction
   frac = SQ_INSTS_VALU / (duration x 1024 SIMDs x 2.4 GHz / 2)
(one wave64 VALU instruction per 2 cycles per SIMD is the issue PEAK; tools/valu_rate_probe.hip prices the mix: plain VOP2 2.6, VOP3 /
packed / DPP 4.3, transcendentals 8.2 cycles - a kernel at 0.5-0.6 of this fraction is issue-bound).

  python tools/forms_summary.py <stats dir> <pmc summary json> <out json>
"""
import csv
import glob
import json
import sys

PEAK = 1024 * 2.4e9 / 2


def main():
    stats_dir, pmc_json, out = sys.argv[1:4]
    dur = {}
    for f in glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
    pmc = json.load(open(pmc_json))
    res = {}
    for k, (ns, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        e = pmc.get(k)
        if not e or "SQ_INSTS_VALU_mean" not in e or ns < 5000:
            continue
        insts = e["SQ_INSTS_VALU_mean"]
        res[k] = dict(avg_us=ns / 1e3, calls=calls, valu_wave_instructions=insts, valu_issue_frac=insts / (ns * 1e-9 * PEAK),
                      sq_busy_cycles=e.get("SQ_BUSY_CYCLES_mean"), sq_active_inst_valu=e.get("SQ_ACTIVE_INST_VALU_mean"))
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res.items():
        print("%-100s %9.1f us  %5.2f of VALU issue peak" % (k[:100], e["avg_us"], e["valu_issue_frac"]))


if __name__ == "__main__":
    main()
