#!/bin/bash
# GPU timeline of the host-fed training loop: kernel + memory-copy trace of bench.py --h2d, then the idle gaps and the copies
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/h2d_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-vendor-gemm --h2d > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.path.join(os.environ.get('GRAFT_REPO_ROOT', os.getcwd()), 'gpurun_out', 'h2d_trace')
k = glob.glob(O + '/**/*kernel_trace.csv', recursive=True)[0]
m = glob.glob(O + '/**/*memory_copy_trace.csv', recursive=True)
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]) for r in csv.DictReader(open(k))]
rows.sort()
t_end = rows[-1][1]
sel = [r for r in rows if r[0] > t_end - 80e6]           # the last 80 ms: steady-state timed steps
busy = sum(e - s for s, e, _ in sel)
span = sel[-1][1] - sel[0][0]
gaps = sorted(((sel[i + 1][0] - sel[i][1], sel[i][2], sel[i + 1][2]) for i in range(len(sel) - 1)), reverse=True)[:8]
print('last 80 ms: kernels busy %.1f ms of %.1f ms span (%d launches)' % (busy / 1e6, span / 1e6, len(sel)))
for g, a, b in gaps:
    print('  idle %.2f ms between %s and %s' % (g / 1e6, a, b))
if m:
    cp = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', ''), r) for r in csv.DictReader(open(m[0]))]
    cp = [c for c in cp if c[0] > t_end - 80e6]
    for s, e, d, r in cp[:12]:
        print('  copy %s start %+.2f ms dur %.3f ms' % (d, (s - sel[0][0]) / 1e6, (e - s) / 1e6))
PY
rm -rf $O/*/
