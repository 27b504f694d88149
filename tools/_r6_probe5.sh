R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O
cd $R

timeout 300 python tools/rank_trace.py 8 150 12 2>&1 | grep cfg4
timeout 300 python tools/rank_trace.py 1 150 12 2>&1 | grep cfg4
timeout 300 python tools/rank_trace.py 4 150 12 2>&1 | grep cfg4 | tail -1
timeout 300 python tools/rank_trace.py 2 150 12 2>&1 | grep cfg4 | tail -1
