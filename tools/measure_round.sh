set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r1f; mkdir -p $O
cd $R
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2>/dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o b -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o b -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write rollout_stream_kernel $O/rollout_traffic.json
python tools/pmc_summary.py $O/pmc_all_kernels.json $O/pmc_fetch $O/pmc_write > /dev/null
python tools/pmc_summary.py $O/pmc_mfma.json $O/pmc_mfma > /dev/null
timeout 600 python tools/configs_bench.py $O/configs.json > $O/configs.log 2>&1
timeout 300 python tools/shard_time.py > $O/shard_time.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
find $O/stats -name "*kernel_trace*" -delete
tail -c 600 $O/bench.json; head -8 $O/stats/b_kernel_stats.csv | cut -c1-150; tail -3 $O/shard_time.txt
