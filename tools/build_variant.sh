# A variant of the product library for A/B runs: ONE translation unit recompiled with extra flags, linked with the product
# build's other objects (python __graft_entry__.py first).
#   bash tools/build_variant.sh <name> <unit.hip> [-DFLAG ...]   ->  scratch_libs/<name>.so
set -e
N=$1; U=$2; shift 2
C=uuo_mocap_amd/csrc
mkdir -p scratch_libs scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fno-gpu-rdc "$@" -c $C/$U -o scratch/variants/${N}_$U.o
OBJS=""
for o in $(python -c "import __graft_entry__ as g; print(' '.join('uuo_mocap_amd/csrc/build/%s.o' % s for s in g.HIP_SOURCES))"); do
  if [ "$(basename $o)" = "$U.o" ]; then OBJS="$OBJS scratch/variants/${N}_$U.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map -o scratch_libs/$N.so $OBJS
echo built scratch_libs/$N.so
