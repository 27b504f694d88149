R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_lists or far_units" 2>&1 | tail -5
for ov in 0 1; do
  for G in 8 1; do
    if [ $ov = 1 ]; then export DUST_OVERLAP=1; else unset DUST_OVERLAP; fi
    timeout 300 python tools/rank_trace.py $G 150 20 2>&1 | grep cfg4 | tail -2 | sed "s/^/overlap=$ov /"
  done
done
export DUST_OVERLAP=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr8 -o b -- python3 $R/tools/rank_trace.py 8 150 4 > /dev/null 2>&1
cd $R
python tools/trace_seq.py $(find /tmp/tr8 -name "*kernel_trace.csv" | head -1) 18 | cut -c1-150
