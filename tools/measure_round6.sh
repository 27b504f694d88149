# Round 6.  One gpurun call: the round's bench lines (driver's command and the default), rocprofv3 kernel stats of an OPEN-LOOP-ONLY run
# (the svmpc_tick2_kernel row is the timed kernel and nothing else - VERDICT r5 item 3) and of the full run, PMC passes (each in its own
# run), derived summaries stamped with the kernel sources they were measured on (tools/srcstamp.py).  Results land in gpurun_out/r6m/ ;
# the summaries to be judged are copied to profiles/round6_* afterwards.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_open -o b -- python3 $R/bench.py --open-loop-only > $O/bench_open_loop_under_rocprof.json 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o b -- python3 $R/bench.py --open-loop-only --steps 40 --warmup 5 > /dev/null 2>&1
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
                   "source": "rocprofv3 --pmc SQ_* --kernel-trace -- python3 bench.py --open-loop-only --steps 40 --warmup 5"},
                  open("$O/tick_pmc.json", "w"), indent=1)
PY
timeout 900 python tools/configs_bench.py $O/configs.json > $O/configs.log 2>&1
timeout 200 python tools/closed_loop_probe.py > $O/closed_loop_probe.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
find $O/stats $O/stats_open -name "*kernel_trace*" -delete
head -5 $O/stats_open/*/b_kernel_stats.csv 2>/dev/null | cut -c1-160 || find $O/stats_open -name "*kernel_stats.csv" | head
# cfg4: the aged set.  Tick time and left-out shares tick by tick; per-kernel durations of the LAST launches of a 150-tick run; the rank's tick
timeout 200 python tools/far_probe.py 150 > $O/far_probe.txt 2>&1
(cd /tmp && DUST_PROBE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_far -o b -- python3 $R/tools/far_probe.py 150 > /dev/null 2>&1)
python tools/trace_tail.py $(find /tmp/tr_far -name "*kernel_trace.csv" | head -1) 10 > $O/far_kernels_aged.txt 2>&1
(cd /tmp && DUST_PROBE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_far2 -o b -- python3 $R/tools/far_probe.py 12 > /dev/null 2>&1)
python tools/trace_tail.py $(find /tmp/tr_far2 -name "*kernel_trace.csv" | head -1) 5 > $O/far_kernels_fresh.txt 2>&1
timeout 300 python tools/far_granularity.py 150 > $O/far_granularity.txt 2>&1
for G in 1 2 4 8; do timeout 300 python tools/rank_trace.py $G 150 12 2>&1 | grep cfg4; done > $O/rank_tick.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr8 -o b -- python3 $R/tools/rank_trace.py 8 150 3 > /dev/null 2>&1)
python tools/trace_seq.py $(find /tmp/tr8 -name "*kernel_trace.csv" | head -1) 20 > $O/rank8_kernel_sequence.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o b -- python3 $R/tools/rank_trace.py 1 150 3 > /dev/null 2>&1)
python tools/trace_seq.py $(find /tmp/tr1 -name "*kernel_trace.csv" | head -1) 20 > $O/rank1_kernel_sequence.txt 2>&1
timeout 300 python tools/shard_emul.py 1,2,4,8 160 2>&1 | grep cfg4 > $O/shard_emul_aged.txt
timeout 300 python tools/peer_probe.py 2,4 > $O/peer_probe.txt 2>&1
bash tools/path_survey.sh > $O/path_survey.txt 2>&1
# cfg2 / K2: rate of the default form and of the launches apart, the steady-state launch sequence, the bandwidth role's own timeline
(python tools/k2_seq.py 300; python tools/k2_seq.py 300; DUST_K2_FORM=0 python tools/k2_seq.py 300) > $O/k2_sequence.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trk2 -o b -- python3 $R/tools/k2_seq.py 60 > /dev/null 2>&1)
python tools/trace_seq.py $(find /tmp/trk2 -name "*kernel_trace.csv" | head -1) 16 >> $O/k2_sequence.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DK2_STAMPS -Iinclude -Idust_amd/csrc tools/k2_bw_probe.hip -o /tmp/k2_bw_probe 2>/dev/null && /tmp/k2_bw_probe >> $O/k2_sequence.txt 2>&1
(python tools/cfg5_seq.py 100; cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trc5 -o b -- python3 $R/tools/cfg5_seq.py 30 > /dev/null 2>&1; cd $R; python tools/trace_seq.py $(find /tmp/trc5 -name "*kernel_trace.csv" | head -1) 14) > $O/cfg5_control_sequence.txt 2>&1
timeout 300 python tools/states_probe.py > $O/states_probe.txt 2>&1
python tools/states_hbm.py $O/states_probe.txt $O/states_hbm.json > /dev/null
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 DUST_BENCH_FORCE_DIST=1 timeout 400 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/sharded_world1_bench.json 2> $O/sharded_world1_bench.err
tail -c 1500 $O/bench_driver_cmd.json; cat $O/rank_tick.txt; tail -c 900 $O/sharded_world1_bench.json
