#!/bin/bash
# A/B builds of libmotioned.so with one source compiled under extra -D flags (ablations, experiments); loaded through ME_LIB=<path>.
#   tools/build_abl.sh <tag> <source.hip> <flags...>      ->  tools/_bin/libmotioned_<tag>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; src=$2; shift 2
mkdir -p $R/tools/_bin
extra=""
[ "$src" = "attn.hip" ] && extra="-ffinite-math-only"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $extra "$@" -c $R/motioneditor_amd/csrc/$src -o $R/tools/_bin/${src%.hip}_$tag.o
objs=""
for o in $R/motioneditor_amd/csrc/_obj/*.o; do
  [ "$(basename $o)" = "${src%.hip}.o" ] && o=$R/tools/_bin/${src%.hip}_$tag.o
  objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_bin/libmotioned_$tag.so $objs
rm -f $R/tools/_bin/${src%.hip}_$tag.o
echo $R/tools/_bin/libmotioned_$tag.so
