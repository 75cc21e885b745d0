tools/r3_var.sh base tc_p1 tc_p2
for v in base twg_p1 twg_p2; do echo "== $v"; ISTGCN_LIB_PATH=tools/bin/lib_$v.so timeout -k 10 120 python tools/kbench.py --only wgrad 2>&1 | grep -v amdgpu.ids; done
