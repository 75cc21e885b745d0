#!/bin/bash
# A/B of tconv experiment builds against the in-tree library (tools/tconv_var_exp.py): usage tools/r3_var12.sh <name>...
timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do ISTGCN_LIB_PATH=tools/bin/lib_$v.so timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids
