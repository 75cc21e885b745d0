#!/bin/bash
# Experiment build: one source compiled with extra flags, linked with the other objects of the regular build into
# tools/bin/lib_<name>.so (travels to the GPU box; selected with ISTGCN_LIB_PATH=tools/bin/lib_<name>.so).
# usage: tools/build_variant.sh <name> <source.hip> [-D...]
name=$1; src=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
B=$R/ist-gcn_amd/build; O=$R/tools/bin/obj_$name; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I $R/ist-gcn_amd/csrc "$@" -c $R/ist-gcn_amd/csrc/$src -o $O/$src.o || exit 1
objs=$(ls $B/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/bin/lib_$name.so $objs $O/$src.o && echo built tools/bin/lib_$name.so
