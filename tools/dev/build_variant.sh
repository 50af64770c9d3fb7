#!/bin/bash
# An alternative library build with extra compiler flags on ONE source:  tools/dev/build_variant.sh <name> <source.hip> <flags...>
#   -> .ab/lib<name>.so (travels with the snapshot; run with SBMC_HIP_LIB=$PWD/.ab/lib<name>.so)
set -e
cd /root/repo
name=$1; src=$2; shift 2
mkdir -p .ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function "$@" -c -o .ab/$name.o sbmc_amd/csrc/$src
objs=$(ls sbmc_amd/.obj/*.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -o .ab/lib$name.so .ab/$name.o $objs
rm .ab/$name.o
echo .ab/lib$name.so
