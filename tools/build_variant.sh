# builds the library from the working tree (or from a git revision given as $2) into $1: A/B timing of two builds on ONE box
set -e
T2F=${T2F--mllvm -disable-machine-licm}  # as __graft_entry__.UNIT_CFLAGS (T2F="" builds with machine LICM)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; REV=$2
SRC=$R
if [ -n "$REV" ]; then
  SRC=/tmp/dust_variant_$$; rm -rf $SRC; mkdir -p $SRC; (cd $R && git archive $REV dust_amd/csrc include) | tar -x -C $SRC
fi
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -I$SRC/include -I$SRC/dust_amd/csrc $T2X"
/opt/rocm/bin/hipcc $F -c $SRC/dust_amd/csrc/dust_amd.hip -o /tmp/variant_a_$$.o &
/opt/rocm/bin/hipcc $F $T2F -c $SRC/dust_amd/csrc/tick2.hip -o /tmp/variant_b_$$.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/variant_a_$$.o /tmp/variant_b_$$.o -ldl -o $OUT
rm -f /tmp/variant_a_$$.o /tmp/variant_b_$$.o
