R=$GRAFT_REPO_ROOT
cd $R
for v in "" "DUST_PACK_ORDER=0" "DUST_PACK_MERGE=0"; do
  echo "== $v"
  env $v timeout 300 python tools/rank_trace.py 2 150 10 2>&1 | grep cfg4 | tail -1
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr2 -o b -- python3 $R/tools/rank_trace.py 2 150 3 > /dev/null 2>&1
cd $R
python tools/trace_seq.py $(find /tmp/tr2 -name "*kernel_trace.csv" | head -1) 22 | cut -c1-160
