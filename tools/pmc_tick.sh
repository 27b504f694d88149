cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/pmc1; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O -o b -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > /dev/null 2>&1
cd $R; python tools/pmc_summary.py $O/../pmc_tick2.json $O > /dev/null; python - <<PY
import json
d=json.load(open("$O/../pmc_tick2.json"))
for k,e in d.items():
    if "tick" in k: print(k[:60], {a:round(b) for a,b in e.items() if a.endswith("_mean")})
PY
