"""profiles/round4_states_probe.txt (tools/states_probe.py's output) -> profiles/round4_states_hbm.json: per stored-states form the
algorithmic bytes, the time and the HBM fraction (bench.py's roofline.forms reads it).  python tools/states_hbm.py <in.txt> <out.json>"""
import json, re, sys
out = {}
for line in open(sys.argv[1]):
    m = re.match(r"(particle|pendulum) N=(\d+) S=(\d+) M=(\d+) H=(\d+) store=(\w+) f16=(\w+): ([\d.]+) us, ([\d.]+) GB algorithmic -> (\d+) GB/s", line)
    if m:
        model, N, S, M, H, store, f16, us, gb, gbs = m.groups()
        key = "%s_%s_%s" % (model, "store" if store == "True" else "nostore", "f16" if f16 == "True" else "f32")
        out[key] = {"model": model, "N": int(N), "S": int(S), "M": int(M), "H": int(H), "total_us": float(us), "algorithmic_gb": float(gb),
                    "achieved_gbs": float(gbs), "hbm_frac": round(float(gbs) / 8000.0, 3)}
# attach the "whole-line" lines: each precedes its form's line
res, pend = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"\s+whole-line states kernel ([\d.]+) us \+ second pass ([\d.]+) us", line)
    if m:
        pend = {"states_kernel_us": float(m.group(1)), "second_pass_us": float(m.group(2))}
        continue
    m = re.match(r"(particle|pendulum) N=\d+ S=\d+ M=\d+ H=\d+ store=(\w+) f16=(\w+):", line)
    if m:
        key = "%s_%s_%s" % (m.group(1), "store" if m.group(2) == "True" else "nostore", "f16" if m.group(3) == "True" else "f32")
        res[key] = dict(out[key], **(pend or {}))
        pend = None
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res, indent=1))
