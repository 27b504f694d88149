# development: product + diagnostic libraries in parallel
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Iinclude -Idust_amd/csrc dust_amd/csrc/dust_amd.hip"
(/opt/rocm/bin/hipcc $F -DDUST_STAMPS $EXTRA -o tools/libdust_amd_stamps.so 2>&1 | grep -E "error|Error") &
(/opt/rocm/bin/hipcc $F $EXTRA -o dust_amd/libdust_amd.so 2>&1 | grep -E "error|Error") &
wait
ls -la dust_amd/libdust_amd.so tools/libdust_amd_stamps.so
