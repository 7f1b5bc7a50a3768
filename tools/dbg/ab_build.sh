#!/bin/bash
# Variant library for an A/B measurement on one box: recompile ONE source with extra -D flags and link it with the product's
# other objects into srslte_amd/lib/ab_<name>.so (select it with SRSRAN_HIP_LIB=...).  usage: ab_build.sh <name> <source.hip | /abs/path/of/an/edited/copy.hip> [-DX ...]
set -e
name=$1; src=$2; shift 2
root=$(cd $(dirname $0)/../.. && pwd)
obj=$root/build/ab/$name.o; mkdir -p $root/build/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$root/include -I$root/srslte_amd/csrc "$@" -x hip -c $(case $src in /*) echo $src;; *) echo $root/srslte_amd/csrc/$src;; esac) -o $obj 2>/dev/null
objs=$(ls $root/build/obj/*.o | grep -v "/$(basename $src).o")
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $root/srslte_amd/lib/ab_$name.so $objs $obj 2>/dev/null
echo built srslte_amd/lib/ab_$name.so
