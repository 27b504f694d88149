timeout 200 python tools/persist_check.py 2>&1 | tail -12
timeout 100 python tools/_hostt.py 2>&1 | tail -2
DUST_AMD_LIB=tools/libdust_amd_stamps.so timeout 200 python tools/tick_timeline.py 2>&1 | grep "k=2\|k=5"
