#!/bin/bash
# (profiles/mkvariant.sh: builds land in build_tmp/ab/, which is scratch and not tracked; profiles/ab_builds.py and profiles/ab_env.py run them)
# usage: mkvar2.sh <name> <tu> [-D...]  -> build_tmp/ab/lib_<name>.so : the working tree's <tu>.hip compiled with the extra flags, linked with the other objects of build/core
set -e
name=$1; tu=$2; shift 2
mkdir -p build_tmp/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-assume "$@" -c -o build_tmp/ab/${tu}_$name.o tungsten_amd/csrc/$tu.hip
objs=$(ls build/core/*.o | grep -v "/$tu.o")
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o build_tmp/ab/lib_$name.so build_tmp/ab/${tu}_$name.o $objs
echo built build_tmp/ab/lib_$name.so
