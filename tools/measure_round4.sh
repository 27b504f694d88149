# Round 4.  One gpurun call: the round's bench lines (driver's command and the default), rocprofv3 kernel stats, PMC passes (each in its own
# run), derived summaries.  Results land in gpurun_out/r4m/ ; the summaries to be judged are copied to profiles/round4_* afterwards.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; mkdir -p $O
cd $R
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o b -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "particle_states_kernel<2>" $O/rollout_states_traffic.json
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "svmpc_tick2_kernel" $O/tick_traffic.json
python tools/pmc_summary.py $O/pmc_sq.json $O/pmc_sq > /dev/null
python - <<PY
import json
d = json.load(open("$O/pmc_sq.json"))
for k, e in d.items():
    if "svmpc_tick2_kernel" in k:
        json.dump({"kernel": k, "SQ_INSTS_VALU_per_tick": e["SQ_INSTS_VALU_mean"], "counters": e,
                   "source": "rocprofv3 --pmc SQ_* --kernel-trace -- python3 bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5"},
                  open("$O/tick_pmc.json", "w"), indent=1)
PY
timeout 900 python tools/configs_bench.py $O/configs.json > $O/configs.log 2>&1
timeout 200 python tools/coldstart.py > $O/coldstart.txt 2>&1
timeout 200 python tools/bench_timing_probe.py > $O/bench_timing_probe.txt 2>&1
timeout 200 python tools/closed_loop_probe.py > $O/closed_loop_probe.txt 2>&1
timeout 100 tools/_valu_rate_probe > $O/valu_rate_probe.txt 2>&1
timeout 100 tools/_ldsdma_probe > $O/ldsdma_probe.txt 2>&1
DUST_AMD_LIB=tools/_libdust_stamps.so timeout 200 python tools/tick2_timeline.py > $O/tick2_timeline.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
find $O/stats -name "*kernel_trace*" -delete
tail -c 1200 $O/bench_driver_cmd.json; head -6 $O/stats/*/b_kernel_stats.csv 2>/dev/null | cut -c1-160 || find $O/stats -name "*kernel_stats.csv" | head
# --- appended: cfg4 sharding projection, world-1 forced-collective bench, pair probe, sparsity, states probe
cd $R
timeout 600 python tools/shard_time.py cfg4 > $O/shard_time.txt 2>&1
timeout 300 python tools/states_probe.py > $O/states_probe.txt 2>&1
python tools/states_hbm.py $O/states_probe.txt $O/states_hbm.json > /dev/null
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 DUST_BENCH_FORCE_DIST=1 timeout 400 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/sharded_world1_bench.json 2> $O/sharded_world1_bench.err
tail -3 $O/shard_time.txt; tail -c 600 $O/sharded_world1_bench.json
# --- round 4: what bounds each stored-states form and the cfg3 / cfg5 rollout launches: kernel durations from a
# --kernel-trace --stats run, VALU instruction counts from a separate --pmc run of the same command, combined by tools/forms_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_stats -o b -- python3 $R/tools/states_probe.py > $O/states_probe_under_rocprof.txt 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/st_pmc -o b -- python3 $R/tools/states_probe.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cf_stats -o b -- python3 $R/tools/configs_bench.py $O/configs_under_rocprof.json > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/cf_pmc -o b -- python3 $R/tools/configs_bench.py /dev/null > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/states_pmc.json $O/st_pmc > /dev/null
python tools/pmc_summary.py $O/configs_pmc.json $O/cf_pmc > /dev/null
python tools/forms_summary.py $O/st_stats $O/states_pmc.json $O/states_forms.json
python tools/forms_summary.py $O/cf_stats $O/configs_pmc.json $O/configs_forms.json
find $O/st_stats $O/cf_stats -name "*kernel_trace*" -delete
rm -rf $O/st_pmc $O/cf_pmc
cat $O/states_forms.json | head -60
# --- K2 (iid_mp) tick: time and kernels
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k2_stats -o b -- python3 $R/tools/k2_time.py > $O/k2_time_under_rocprof.txt 2>&1
cd $R
timeout 200 python tools/k2_time.py > $O/k2_time.txt 2>&1
find $O/k2_stats -name "*kernel_trace*" -delete
