# builds only the tick2 translation unit with extra flags ($2...) and links it against the current dust_amd.o into $1
# (A/B timing of tick-kernel variants on ONE box: python tools/ab_time.py libA.so libB.so)
set -e
T2F=${T2F--mllvm -disable-machine-licm}  # as __graft_entry__.UNIT_CFLAGS (T2F="" builds with machine LICM)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -I$R/include -I$R/dust_amd/csrc $*"
T=$(mktemp /tmp/t2v_XXXXXX.o)
/opt/rocm/bin/hipcc $F $T2F -c $R/dust_amd/csrc/tick2.hip -o $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/dust_amd/build/dust_amd.o $T -ldl -o $OUT
rm -f $T
