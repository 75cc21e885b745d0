#!/bin/bash
# graph conv weight gradient per layer shape: in-tree vs experiment builds; usage: tools/ab_gwg.sh [-r reps] variant...
reps=2; [ "$1" = "-r" ] && { reps=$2; shift 2; }
L=${GWG_LAYERS:-64x64x300,64x128x300,128x128x150,128x256x150,256x256x75}
for rep in $(seq 1 $reps); do
  echo "== in-tree (rep $rep)"; python tools/kbench.py --only gcn_wgrad --layers $L
  for v in "$@"; do echo "== $v (rep $rep)"; ISTGCN_LIB_PATH=tools/bin/lib_$v.so python tools/kbench.py --only gcn_wgrad --layers $L; done
done
