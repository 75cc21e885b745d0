#!/bin/bash
# The clock the chip holds under load: GRBM_GUI_ACTIVE (cycles the graphics engine was busy, at the shader clock) per kernel,
# divided by the kernel's duration, for the vendor GEMM on all-zero / random operands and for one training step's kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clock_probe; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/gemm -- python3 $R/tools/gemm_peak.py > $O/gemm.log 2>&1 || { tail -5 $O/gemm.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/step -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vendor-gemm > $O/step.log 2>&1 || { tail -5 $O/step.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ.get('GRAFT_REPO_ROOT', os.getcwd()), 'gpurun_out', 'clock_probe')
for sub in ('gemm', 'step'):
    cc = glob.glob(O + '/%s/**/*counter_collection.csv' % sub, recursive=True)
    kt = glob.glob(O + '/%s/**/*kernel_trace.csv' % sub, recursive=True)
    if not cc or not kt:
        print(sub, 'missing csv', cc, kt); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
    rows = list(csv.DictReader(open(cc[0])))
    if sub == 'gemm':
        print('columns:', list(rows[0].keys()))
    agg = collections.OrderedDict()
    for r in rows:
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE':
            continue
        d = dur.get(r['Dispatch_Id'])
        if not d or d[0] < 20000:
            continue
        name = d[1].replace('(anonymous namespace)::', '')[:48]
        a = agg.setdefault(name, [0.0, 0.0, 0])
        a[0] += float(r['Counter_Value']); a[1] += d[0]; a[2] += 1
    print('== %s: cycles / ns per kernel family (launches >= 20 us)' % sub)
    for name, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print('%-50s n=%4d  avg %8.1f us  GRBM_GUI_ACTIVE / ns = %.3f' % (name, n, ns / n / 1e3, cyc / ns))
PY
rm -rf $O/gemm/*/ $O/step/*/ 2>/dev/null
