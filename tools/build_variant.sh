#!/bin/bash
# Experiment build: one or more sources (comma separated) compiled with extra flags, linked with the other objects of the
# regular build into tools/bin/lib_<name>.so (travels to the GPU box; selected with ISTGCN_LIB_PATH=tools/bin/lib_<name>.so).
# usage: tools/build_variant.sh <name> <source.hip[,source2.hip...]> [-D...]
name=$1; srcs=${2//,/ }; shift 2
R=$(cd $(dirname $0)/.. && pwd)
B=$R/ist-gcn_amd/build; O=$R/tools/bin/obj_$name; mkdir -p $O
pids=()
for src in $srcs; do
  own=$(grep -m1 -E '^//[[:space:]]*hipcc-flags:' $R/ist-gcn_amd/csrc/$src | sed -E 's/^.*hipcc-flags:[[:space:]]*//')     # the source's own flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I $R/ist-gcn_amd/csrc $own "$@" -c $R/ist-gcn_amd/csrc/$src -o $O/$src.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
objs=$(ls $B/*.o)
for src in $srcs; do objs=$(echo "$objs" | grep -v "/$src.o"); done
mine=$(for src in $srcs; do echo $O/$src.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/bin/lib_$name.so $objs $mine && echo built tools/bin/lib_$name.so
