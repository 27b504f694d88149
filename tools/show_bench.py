"""Prints the fields of a bench.py JSON line that matter at a glance."""
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "warmup_ticks_run", "closed_loop_ticks_per_s", "tick_paths", "scale_workload")})
r = d.get("roofline") or {}
print({k: r.get(k) for k in ("frac", "avg_launch_us", "launch_us_median", "launch_us_p90", "frac_at_median", "traffic")})
print(r.get("product_kernel"))
print((d.get("cpu_baseline") or {}).get("value"))
