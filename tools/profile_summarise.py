"""Condense the rocprofv3 passes of tools/profile_bench.sh into the files kept under profiles/:
  <tag>_kernel_stats.csv   per-kernel calls / total / average duration (the --stats table)
  <tag>_pmc.json           per kernel family: HBM bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
                           gfx950, WRITE_SIZE exact; rocprofv3 reports KB) and the SQ counters with the derived MFMA
                           utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32): SQ_BUSY_CYCLES comes out summed
                           over the 32 shader engines, each with 32 SIMDs, the MFMA counter over all 1024 SIMDs."""
import collections, csv, glob, json, os, shutil, sys
O, tag = sys.argv[1], sys.argv[2]
BENCH_ARGS = ' '.join(sys.argv[3:])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')


def fam(name):
    n = name.replace('(anonymous namespace)::', '')
    if 'bneck_wgrad' in n:
        return 'bneck_wgrad kernels'
    if 'bneck_in_kernel' in n or 'bneck_out_kernel' in n:
        return 'bneck_in/out kernels'
    if 'tconv_lean_kernel' in n or 'tconv_kernel' in n:      # (one C-ABI entry point, istgcn_tconv, dispatches to both)
        return 'tconv kernels'
    if 'twg_lean_kernel' in n or 'twg_ws_kernel' in n or 'tconv_rc_wgrad_kernel' in n or 'dz_colsum_kernel' in n:
        return 'tconv_wgrad kernels'                          # (istgcn_tconv_wgrad: lean, round-2 and frame-tiled variants)
    for key in ('gcn_rc_fwd_kernel', 'gcn_rc_bwd_kernel', 'gcn_rc_wgrad_kernel', 'gwg_ws_kernel', 'gcn_bwd_ws_kernel', 'tconv_wgrad_kernel', 'tconv_kernel', 'gcn_fwd_kernel', 'gcn_bwd_kernel', 'wgrad_reduce_kernel', 'block_out_fwd_kernel',
                'block_out_bwd_kernel', 'affine2_kernel', 'bn_finalize_kernel', 'bn_bwd_coef_kernel', 'fold_fwd_kernel', 'fold_bwd_kernel',
                'sgd_step_kernel', 'pool_fwd_kernel', 'pool_bwd_kernel', 'tcn_fold_fwd_kernel', 'tcn_fold_bwd_kernel', 'input_stats_kernel', 'input_apply_kernel', 'input_bwd_kernel', 'pack_'):
        if key in n:
            return key if key != 'pack_' else 'pack kernels'
    return 'other (framework)'


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(O, sub, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = fam(row['Kernel_Name'])
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
            n[k].add(row['Dispatch_Id'])
    return acc, {k: len(v) for k, v in n.items()}


stats = glob.glob(os.path.join(O, 'trace', '**', '*kernel_stats.csv'), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(OUT, '%s_kernel_stats.csv' % tag))
fetch, nf = counters('fetch')
write, nw = counters('write')
sq, ns = counters('sq')
sys.path.insert(0, ROOT)
import importlib.util
_spec = importlib.util.spec_from_file_location('istgcn_lib', os.path.join(ROOT, 'ist-gcn_amd', '_lib.py'))
_lib = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_lib)
res = {'csrc_hash': _lib.csrc_hash(),       # the kernel sources these counters were collected on (== istgcn_build_id() of the library)
       'command': 'rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_*} -- python3 bench.py --steps 4 --warmup 2 '
                  '--no-cpu-baseline %s (four separate passes)' % BENCH_ARGS,
       'units': 'FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; hbm_bytes = 2*FETCH*1024 + WRITE*1024 per launch',
       'kernels': {}}
for k in sorted(set(fetch) | set(write) | set(sq)):
    r = {}
    if k in fetch and k in write:
        f_ = fetch[k]['FETCH_SIZE'] / max(nf[k], 1)
        w_ = write[k]['WRITE_SIZE'] / max(nw[k], 1)
        r.update(launches=nf[k], fetch_kb_raw_avg=round(f_, 1), write_kb_avg=round(w_, 1), hbm_bytes_avg=int(2 * f_ * 1024 + w_ * 1024))
    if k in sq:
        c = sq[k]
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        r.update(mfma_util=round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('SQ_BUSY_CYCLES', 0) * 32, 1), 4),
                 wave_cycles_parked=round(c.get('SQ_WAIT_ANY', 0) / wc, 3), wave_cycles_issue_stalled=round(c.get('SQ_WAIT_INST_ANY', 0) / wc, 3),
                 wave_cycles_issuing=round(c.get('SQ_ACTIVE_INST_ANY', 0) / wc, 3), mfma_instructions=int(c.get('SQ_INSTS_MFMA', 0)))
    res['kernels'][k] = r
# whole-step HBM traffic: every kernel's counter bytes x its launches, over the steps the profiled command ran (bench.py
# --steps 4 --warmup 2: six steps; set-up launches -- packs, the optimizer's first layout -- are in, i.e. an upper bound)
STEPS = 6
tot_b = sum(v.get('hbm_bytes_avg', 0) * v.get('launches', 0) for v in res['kernels'].values())
res['steps_profiled'] = STEPS
res['hbm_bytes_per_step'] = int(tot_b / STEPS)
res['hbm_bytes_per_step_by_kernel'] = {k: int(v.get('hbm_bytes_avg', 0) * v.get('launches', 0) / STEPS) for k, v in res['kernels'].items()
                                       if v.get('hbm_bytes_avg', 0) * v.get('launches', 0) / STEPS > 1e6}
json.dump(res, open(os.path.join(OUT, '%s_pmc.json' % tag), 'w'), indent=1)
print(json.dumps({k: {x: v[x] for x in ('hbm_bytes_avg', 'mfma_util') if x in v} for k, v in res['kernels'].items()}, indent=0))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('total kernel time %.2f ms over %d launches' % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
    for r in rows[:22]:
        print('%6d %9.1f us avg %5.1f%%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage']), r['Name'].replace('(anonymous namespace)::', '')[:100]))
