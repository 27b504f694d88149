# Round 5.  One gpurun call: the round's bench lines (driver's command and the default), rocprofv3 kernel stats, PMC passes (each in its own
# run), derived summaries - stamped with the kernel sources they were measured on (tools/srcstamp.py).  Results land in gpurun_out/r5m/ ;
# the summaries to be judged are copied to profiles/round5_* afterwards.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O
cd $R
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o b -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "particle_states_kernel<2>" $O/rollout_states_traffic.json states
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "svmpc_tick2_kernel" $O/tick_traffic.json tick2
python tools/pmc_summary.py $O/pmc_sq.json $O/pmc_sq > /dev/null
python - <<PY
import json, sys
sys.path.insert(0, "$R/tools")
import srcstamp
d = json.load(open("$O/pmc_sq.json"))
for k, e in d.items():
    if "svmpc_tick2_kernel" in k:
        json.dump({"kernel": k, "SQ_INSTS_VALU_per_tick": e["SQ_INSTS_VALU_mean"], "counters": e, "source_family": "tick2", "source_stamp": srcstamp.stamp("tick2"),
                   "source": "rocprofv3 --pmc SQ_* --kernel-trace -- python3 bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5"},
                  open("$O/tick_pmc.json", "w"), indent=1)
PY
timeout 900 python tools/configs_bench.py $O/configs.json > $O/configs.log 2>&1
timeout 200 python tools/closed_loop_probe.py > $O/closed_loop_probe.txt 2>&1
DUST_AMD_LIB=tools/_libdust_stamps.so timeout 200 python tools/tick2_timeline.py > $O/tick2_timeline.txt 2>&1
DUST_AMD_LIB=tools/_libdust_stamps.so timeout 200 python tools/serve_timeline.py 2000 > $O/serve_timeline.txt 2>&1
timeout 100 tools/_mailbox_probe 256 20000 0 > $O/mailbox_probe.txt 2>&1
timeout 100 tools/_mailbox_probe 256 20000 1 >> $O/mailbox_probe.txt 2>&1
timeout 100 tools/_mailbox_probe 1 20000 0 >> $O/mailbox_probe.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
find $O/stats -name "*kernel_trace*" -delete
tail -c 1500 $O/bench_driver_cmd.json; head -6 $O/stats/*/b_kernel_stats.csv 2>/dev/null | cut -c1-160 || find $O/stats -name "*kernel_stats.csv" | head
cd $R
# pairwise_far.hpp on cfg4: tick time and left-out shares tick by tick (fresh set -> the aged set the bench times), with and without it;
# per-kernel durations of the LAST 10 launches of a 150-tick run (the aged state); one rank's compute with the real data flow
timeout 200 python tools/far_probe.py 150 > $O/far_probe.txt 2>&1
(cd /tmp && DUST_PROBE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_far -o b -- python3 $R/tools/far_probe.py 150 > /dev/null 2>&1)
python tools/trace_tail.py $(find /tmp/tr_far -name "*kernel_trace.csv" | head -1) 10 > $O/far_kernels_aged.txt 2>&1
(cd /tmp && DUST_PROBE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_far2 -o b -- python3 $R/tools/far_probe.py 12 > /dev/null 2>&1)
python tools/trace_tail.py $(find /tmp/tr_far2 -name "*kernel_trace.csv" | head -1) 5 > $O/far_kernels_fresh.txt 2>&1
timeout 200 python tools/far_granularity.py 140 > $O/far_granularity.txt 2>&1
timeout 300 python tools/shard_emul.py 1,2,4,8 > $O/shard_emul.txt 2>&1
DUST_FAR=0 timeout 300 python tools/shard_emul.py 1,8 > $O/shard_emul_far0.txt 2>&1
timeout 300 python tools/shard_emul.py 1,2,4,8 160 2>&1 | grep cfg4 > $O/shard_emul_aged.txt
DUST_AMD_LIB=tools/_libdust_stamps.so timeout 100 python tools/rollout_phases.py > $O/rollout_phases.txt 2>&1
timeout 600 python tools/shard_time.py cfg4 > $O/shard_time.txt 2>&1
timeout 300 python tools/states_probe.py > $O/states_probe.txt 2>&1
python tools/states_hbm.py $O/states_probe.txt $O/states_hbm.json > /dev/null
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 DUST_BENCH_FORCE_DIST=1 timeout 400 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/sharded_world1_bench.json 2> $O/sharded_world1_bench.err
tail -3 $O/shard_time.txt; tail -c 600 $O/sharded_world1_bench.json
