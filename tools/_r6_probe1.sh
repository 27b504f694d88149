set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; mkdir -p $O
cd $R
timeout 300 python tools/rank_trace.py 8 150 20 > $O/rank8.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr8 -o b -- python3 $R/tools/rank_trace.py 8 150 4 > $O/rank8_prof.txt 2>&1
cd $R
python tools/trace_seq.py $(find /tmp/tr8 -name "*kernel_trace.csv" | head -1) 60 > $O/rank8_seq.txt 2>&1
timeout 300 python tools/rank_trace.py 1 150 10 > $O/rank1.txt 2>&1
cat $O/rank8.txt $O/rank1.txt; tail -40 $O/rank8_seq.txt
