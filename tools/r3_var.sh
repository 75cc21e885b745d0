for v in "$@"; do ISTGCN_LIB_PATH=tools/bin/lib_$v.so timeout -k 10 120 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids; done
