R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
timeout 200 python tools/cfg5_loop.py 300 2>&1 | tail -1
MPF=0 timeout 200 python tools/cfg5_loop.py 300 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5 -o b -- python3 $R/tools/cfg5_loop.py 100 > /dev/null 2>&1
cd $R
head -14 $(find /tmp/c5 -name "*kernel_stats.csv" | head -1) | cut -c1-170
python tools/trace_seq.py $(find /tmp/c5 -name "*kernel_trace.csv" | head -1) 8 | cut -c1-170
