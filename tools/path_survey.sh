# Which kernels serve each BASELINE configuration by default: one rocprofv3 kernel-stats run per row of tools/configs_bench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/paths; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for row in "cfg1" "cfg2 (bench.py)" "cfg2 / K2 (iid_mp)" "cfg2 / IMQ" "cfg3" "cfg3 noisy (control-channel noise 0.1)" "cfg3 velocity control" "cfg4 on one GPU" "cfg5 on one GPU (K1, M=8, MPF 256 x 20)"; do
  tag=$(echo "$row" | tr -c 'a-zA-Z0-9' '_')
  rm -rf /tmp/ps_$tag
  DUST_CONFIGS_ONLY="$row" DUST_CONFIGS_EXACT=1 DUST_CONFIGS_NO_WARM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$tag -o b -- python3 $R/tools/configs_bench.py > /tmp/ps_$tag.log 2>&1
  echo "== $row"
  f=$(find /tmp/ps_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("   %6d calls %9.1f us avg %5.1f %%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:100]))
PY
done
