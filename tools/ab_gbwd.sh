#!/bin/bash
# A/B of the graph conv's data gradient (with dA) per layer shape: in-tree library vs experiment builds (tools/bin/lib_<name>.so)
# usage: tools/ab_gbwd.sh <tag> <variant names...>
tag=$1; shift
L=64x64x300,64x128x300,128x128x150,128x256x150,256x256x75
for rep in 1 2; do
  echo "== in-tree (rep $rep)"; python tools/kbench.py --only gcn_bwd_data --layers $L
  for v in "$@"; do echo "== $v (rep $rep)"; ISTGCN_LIB_PATH=tools/bin/lib_$v.so python tools/kbench.py --only gcn_bwd_data --layers $L; done
done
