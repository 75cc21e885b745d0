#!/bin/bash
# Register / spill / occupancy table of every kernel of one source (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kres.sh <source.hip> [extra flags] | c++filt-free: prints the mangled names' template arguments
src=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
own=$(grep -m1 -E '^//[[:space:]]*hipcc-flags:' $R/ist-gcn_amd/csrc/$src | sed -E 's/^.*hipcc-flags:[[:space:]]*//')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I $R/ist-gcn_amd/csrc $own "$@" -c $R/ist-gcn_amd/csrc/$src -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys
cur=None; rec={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rec[cur]={}
    for k in ('VGPRs','AGPRs','VGPRs Spill','SGPRs Spill','Occupancy \[waves/SIMD\]','ScratchSize \[bytes/lane\]'):
        m=re.search(r'^\s*remark:.*?\s'+k+r': (\d+)',l) or re.search(r'\s'+k+r': (\d+)',l)
        if m and cur and k.replace('\\\\','') not in rec[cur]: rec[cur][k.replace('\\\\','')]=int(m.group(1))
for k,v in rec.items():
    print('%-110s vgpr %3d agpr %3d spill %3d scratch %4d occ %d'%(k[:110],v.get('VGPRs',-1),v.get('AGPRs',-1),v.get('VGPRs Spill',-1),v.get('ScratchSize [bytes/lane]',-1),v.get('Occupancy [waves/SIMD]',-1)))
"
rm -f /tmp/kres_$$.o
