"""List the s_waitcnt vmcnt(N) of one kernel with the number of global loads/stores issued so far and the barriers, to
spot waits that are more conservative than the prefetch depth intends (path-dependent load counts)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [m.group(1) for m in re.finditer(r'^(\S+):\s*; @', s, re.M) if pat in m.group(1)]
for nm in names[: int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    i = s.index(nm + ':')
    k = s[i:s.index('.end_amdhsa_kernel', i)]
    lines = [l.strip() for l in k.split('\n') if l.strip() and not l.strip().startswith(';')]
    print('==', nm)
    n_ld = n_st = 0
    for n, l in enumerate(lines):
        if re.match(r'^\.LBB\d+_\d+:', l) and 'Loop' in l:
            print('%5d %s' % (n, l[:70]))
        if 'global_load' in l or 'buffer_load' in l: n_ld += 1
        if 'global_store' in l or 'buffer_store' in l: n_st += 1
        if 'vmcnt' in l:
            print('%5d   %-40s ld %d st %d' % (n, l.split(';')[0].strip(), n_ld, n_st))
        if 's_barrier' in l:
            print('%5d   barrier' % n)
