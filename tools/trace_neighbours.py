"""Kernel trace of one bench command: what runs right before each launch of a named kernel (who launches the copies?).
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...; trace_neighbours.py <dir> copyBuffer"""
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2]
short = lambda n: re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+', '', n)[:60]
prev, nxt, sizes = collections.Counter(), collections.Counter(), collections.Counter()
for i, r in enumerate(rows):
    if pat in r['Kernel_Name']:
        prev[short(rows[i - 1]['Kernel_Name'])] += 1
        if i + 1 < len(rows):
            nxt[short(rows[i + 1]['Kernel_Name'])] += 1
        sizes[(r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')))] += 1
print('launches of', pat, sum(prev.values()))
print('before:', prev.most_common(12))
print('after :', nxt.most_common(12))
print('grids :', sizes.most_common(12))
