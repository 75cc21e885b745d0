#!/bin/bash
mkdir -p gpurun_out
ISTGCN_LIB_PATH=tools/bin/lib_r5base.so timeout -k 10 200 python tools/gwg_bits.py > gpurun_out/gwg_bits_base.txt 2>&1 || { tail -5 gpurun_out/gwg_bits_base.txt; exit 1; }
timeout -k 10 200 python tools/gwg_bits.py > gpurun_out/gwg_bits_new.txt 2>&1 || { tail -20 gpurun_out/gwg_bits_new.txt; exit 1; }
if diff <(grep -v amdgpu.ids gpurun_out/gwg_bits_base.txt) <(grep -v amdgpu.ids gpurun_out/gwg_bits_new.txt) > gpurun_out/gwg_bits_diff.txt; then echo "BIT-IDENTICAL over $(grep -c rel-err gpurun_out/gwg_bits_new.txt) cases"; else echo "DIFFERENT"; head -20 gpurun_out/gwg_bits_diff.txt; fi
grep -c "identical True" gpurun_out/gwg_bits_new.txt; sort -t' ' -k13 gpurun_out/gwg_bits_new.txt | tail -2
L=64x64x300,64x128x300,128x128x150,128x256x150,256x256x75
for rep in 1 2; do
  echo "== r5base (rep $rep)"; ISTGCN_LIB_PATH=tools/bin/lib_r5base.so python tools/kbench.py --only gcn_wgrad --layers $L 2>&1 | grep -v amdgpu.ids
  echo "== new (rep $rep)"; python tools/kbench.py --only gcn_wgrad --layers $L 2>&1 | grep -v amdgpu.ids
done
