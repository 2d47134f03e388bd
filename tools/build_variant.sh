#!/bin/bash
# tools/build_variant.sh NAME SOURCE.hip "-DFLAG ..." : a copy of the package under tools/_var/NAME/crossscore_amd whose library is the in-tree
# objects (crossscore_amd/build/*.o, built by `python -m crossscore_amd.build`) with SOURCE.hip recompiled with the extra flags.  Built HERE
# (hipcc cross-compiles gfx950) so that a gpurun call only runs; tools/_var is git-ignored and travels with the snapshot.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; SRC=$2; shift 2
D=$R/tools/_var/$N
rm -rf "$D"; mkdir -p "$D/crossscore_amd"
cp $R/crossscore_amd/*.py "$D/crossscore_amd/"
cp -r $R/crossscore_amd/config "$D/crossscore_amd/" 2>/dev/null || true
OBJS=""
for o in $R/crossscore_amd/build/*.o; do b=$(basename $o .o); [ "$b.hip" = "$SRC" ] || OBJS="$OBJS $o"; done
EXTRA=""; case $SRC in panel.hip|panel4.hip) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $EXTRA "$@" -c $R/crossscore_amd/csrc/$SRC -o $D/var.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/crossscore_amd/libcrossscore_hip.so $OBJS $D/var.o
rm $D/var.o
echo "$D"
