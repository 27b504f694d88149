# One gpurun call: the round's bench line, rocprofv3 kernel stats, PMC passes (each in its own run), derived summaries.
# Results land in gpurun_out/r2/ ; the summaries to be judged are copied to profiles/round2_* by hand afterwards.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2>/dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o b -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o b -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "particle_states_kernel<2>" $O/rollout_states_traffic.json
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "rollout_stream_kernel<0, false, true" $O/rollout_cfg2_traffic.json
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "svmpc_tick_kernel" $O/tick_traffic.json
python tools/pmc_summary.py $O/pmc_all_kernels.json $O/pmc_fetch $O/pmc_write > /dev/null
python tools/pmc_summary.py $O/pmc_sq.json $O/pmc_sq $O/pmc_mfma > /dev/null
python - <<PY
import json
d = json.load(open("$O/pmc_sq.json"))
for k, e in d.items():
    if "svmpc_tick_kernel" in k:
        json.dump({"kernel": k, "SQ_INSTS_VALU_per_tick": e["SQ_INSTS_VALU_mean"], "counters": e,
                   "source": "rocprofv3 --pmc SQ_* --kernel-trace -- python3 bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5"},
                  open("$O/tick_pmc.json", "w"), indent=1)
PY
timeout 900 python tools/configs_bench.py $O/configs.json > $O/configs.log 2>&1
# large-set pairwise (cfg4 shape on one GPU): kernel durations and the matrix-core counters of the Gram x score GEMM
timeout 300 python tools/pair_probe.py > $O/pair_probe.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pair_stats -o p -- python3 $R/tools/pair_probe.py cfg4 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pair_pmc -o p -- python3 $R/tools/pair_probe.py cfg4 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pair_pmc.json $O/pair_pmc > /dev/null
rm -rf $O/pair_pmc; find $O/pair_stats -name "*kernel_trace*" -delete
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_mfma
find $O/stats -name "*kernel_trace*" -delete
tail -c 1500 $O/bench.json; head -8 $O/stats/b_kernel_stats.csv | cut -c1-150; cat $O/rollout_states_traffic.json
