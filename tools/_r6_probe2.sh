set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused or far or large or zero_blocks or full_size or deterministic" > $O/tests.txt 2>&1
tail -15 $O/tests.txt
timeout 300 python tools/rank_trace.py 8 150 20 > $O/rank8.txt 2>&1
timeout 300 python tools/rank_trace.py 1 150 10 > $O/rank1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr8 -o b -- python3 $R/tools/rank_trace.py 8 150 4 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o b -- python3 $R/tools/rank_trace.py 1 150 4 > /dev/null 2>&1
cd $R
python tools/trace_seq.py $(find /tmp/tr8 -name "*kernel_trace.csv" | head -1) 36 > $O/rank8_seq.txt 2>&1
python tools/trace_seq.py $(find /tmp/tr1 -name "*kernel_trace.csv" | head -1) 36 > $O/rank1_seq.txt 2>&1
cat $O/rank8.txt $O/rank1.txt; tail -20 $O/rank8_seq.txt; tail -20 $O/rank1_seq.txt
