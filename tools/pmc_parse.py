"""Summarise a rocprofv3 --pmc csv directory: per kernel, the summed SQ counters and the derived ratios
(MFMA busy share of the busy cycles, share of wave-cycles parked in s_waitcnt/barrier, issue-stalled, issuing)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for f in files:
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        k = k.replace('(anonymous namespace)::', '')[:110]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        key = (row['Dispatch_Id'], k)
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
    if 'kernel' not in k:
        continue
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    busy = c.get('SQ_BUSY_CYCLES', 0) or 1
    print('%-72s n=%d' % (k, cnt[k]))
    print('    mfma_busy/busy_cycles(per-SE sum) %.3f   wait_any %.2f  wait_inst_any %.2f  active_inst %.2f  wait_inst_lds %.3f  (of wave cycles);  mfma insts %d' % (
        c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc,
        c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_WAIT_INST_LDS', 0) / wc, c.get('SQ_INSTS_MFMA', 0)))
    n_ = max(cnt[k], 1)
    print('    per dispatch: VALU %.3g SALU %.3g LDS %.3g VMEM_RD %.3g VMEM_WR %.3g MFMA %.3g | lds_busy %.2f of busy cycles, conflict share of lds cycles %.2f' % (
        c.get('SQ_INSTS_VALU', 0) / n_, c.get('SQ_INSTS_SALU', 0) / n_, c.get('SQ_INSTS_LDS', 0) / n_, c.get('SQ_INSTS_VMEM_RD', 0) / n_,
        c.get('SQ_INSTS_VMEM_WR', 0) / n_, c.get('SQ_INSTS_MFMA', 0) / n_, c.get('SQ_LDS_IDX_ACTIVE', 0) / busy,
        c.get('SQ_LDS_BANK_CONFLICT', 0) / (c.get('SQ_LDS_IDX_ACTIVE', 0) or 1)))
    print('    raw: ' + ' '.join('%s=%.4g' % kv for kv in sorted(c.items())))
