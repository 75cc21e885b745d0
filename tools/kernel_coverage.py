"""Which kernel of the library is reached by what?  Joins
  * the kernel symbols registered in ist-gcn_amd/libistgcn_hip.so (every instantiation the library was built with),
  * gpurun_out/kernel_coverage.tsv      -- per `-m gpu` test, the kernels it launched (tests/conftest.py, istgcn_trace),
  * gpurun_out/kernel_coverage_bench.tsv -- per BASELINE configuration / storage type, the kernels a bench.py run launched
    (ISTGCN_TRACE_KERNELS=<file> python bench.py ...),
into one markdown table per kernel template: instantiations built / launched by the suite / launched by the bench
configurations, the configurations that reach it, and up to three tests that run it.  A template with instantiations but no
launches is dead weight or an untested dispatch branch.  usage: python tools/kernel_coverage.py > profiles/rNN_kernel_coverage.md"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'ist-gcn_amd', 'libistgcn_hip.so')


def base(n):
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', n) or re.match(r'_Z(\d+)', n)
    if not m:
        return n
    k = int(m.group(1))
    return n[len(m.group(0)):][:k]


out = subprocess.run(['strings', '-n', '8', LIB], stdout=subprocess.PIPE).stdout.decode()
names = sorted({l for l in out.splitlines() if re.match(r'^_Z[A-Za-z0-9_]*kernel[A-Za-z0-9_]*$', l) and '__device_stub__' not in l})
built = collections.defaultdict(set)
for n in names:
    built[base(n)].add(n)


def read(path):
    by_k, who = collections.defaultdict(set), collections.defaultdict(set)
    if os.path.exists(path):
        for l in open(path):
            t, c, k = l.rstrip('\n').split('\t')
            by_k[base(k)].add(k)
            who[base(k)].add(t)
    return by_k, who


tk, tw = read(os.path.join(ROOT, 'gpurun_out', 'kernel_coverage.tsv'))
bk, bw = read(os.path.join(ROOT, 'gpurun_out', 'kernel_coverage_bench.tsv'))
print('| kernel template | instantiations built | launched by the GPU suite | launched by bench configs | bench configurations | tests (up to 3 of n) |')
print('|---|---|---|---|---|---|')
for b in sorted(built, key=lambda b: (-(len(bk.get(b, ())) > 0), -len(built[b]))):
    # reference-pinned (golden) tests first, then whole-model tests, then the rest
    def rank(t):
        return (0 if 'golden' in t else 1 if 'model' in t or 'reference' in t else 2, t)
    tests = sorted({t.split('::')[0].replace('tests/', '') + '::' + t.split('::')[1].split('[')[0] + (' ' + t.split('] ')[1] if '] {' in t else '')
                    for t in tw.get(b, ())}, key=rank)
    cfgs = sorted({t[len('bench:'):] for t in bw.get(b, ())})
    print('| `%s` | %d | %d | %d | %s | %s |' % (b, len(built[b]), len(tk.get(b, ())), len(bk.get(b, ())), ', '.join(cfgs) or '--',
                                                 (', '.join(tests[:3]) + (' (of %d)' % len(tests) if len(tests) > 3 else '')) or '**none**'))
dead = [b for b in built if not tk.get(b) and not bk.get(b)]
print('\nnever launched by the suite or the bench configurations: %s' % (', '.join('`%s`' % d for d in sorted(dead)) or 'none'))
