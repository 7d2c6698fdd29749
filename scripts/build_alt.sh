#!/bin/bash
# An alternative build of ONE source of the library (timing ablations / A-B of a kernel):
#   scripts/build_alt.sh <name> <source.hip> <extra hipcc flags...>
# -> snap_amd/lib/alt_<name>/libsnap_hip.so (the other objects are the default build's);
# select it at run time with SNAP_HIP_LIB=<path>.
set -eu
cd "$(dirname "$0")/../snap_amd/csrc"
name=$1; src=$2; shift 2
make -j16 >/dev/null
out=../lib/alt_$name
mkdir -p $out
obj=$out/$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function \
  -ffp-contract=off -fno-slp-vectorize "$@" -c $src -o $obj
objs=$(ls ../lib/obj/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o $out/libsnap_hip.so
echo built $out/libsnap_hip.so
