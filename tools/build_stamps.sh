# diagnostic build of the library with in-kernel phase stamps: tools/_libdust_stamps.so (use with DUST_AMD_LIB=...)
set -e
T2F=${T2F--mllvm -disable-machine-licm}  # as __graft_entry__.UNIT_CFLAGS (T2F="" builds with machine LICM)
R=$(cd "$(dirname "$0")/.." && pwd)
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -I$R/include -I$R/dust_amd/csrc -DDUST_STAMPS $T2X"
mkdir -p $R/dust_amd/build
/opt/rocm/bin/hipcc $F -c $R/dust_amd/csrc/dust_amd.hip -o $R/dust_amd/build/stamps_a.o &
/opt/rocm/bin/hipcc $F $T2F -c $R/dust_amd/csrc/tick2.hip -o $R/dust_amd/build/stamps_b.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/dust_amd/build/stamps_a.o $R/dust_amd/build/stamps_b.o -ldl -o $R/tools/_libdust_stamps.so
