#!/bin/bash
# forward graph conv per layer shape: in-tree vs experiment builds
L=64x64x300,64x128x300,128x128x150,128x256x150,256x256x75
for rep in 1 2; do
  echo "== in-tree (rep $rep)"; python tools/kbench.py --only gcn_fwd --layers $L
  for v in "$@"; do echo "== $v (rep $rep)"; ISTGCN_LIB_PATH=tools/bin/lib_$v.so python tools/kbench.py --only gcn_fwd --layers $L; done
done
