"""Basic-block instruction mix of one kernel in a hipcc -save-temps .s file (which blocks carry the instructions).
usage: isa_blocks.py file.s <kernel-name-substring> [min_instructions] [dump-label]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dump = sys.argv[4] if len(sys.argv) > 4 else None
names = [m.group(1) for m in re.finditer(r'^([A-Za-z_0-9]*%s[A-Za-z_0-9]*):' % re.escape(pat), s, re.M)]
name = names[0]
i = s.index(name + ':')
j = s.index('.end_amdhsa_kernel', i)
blocks, cur = [], ['entry', []]
blocks.append(cur)
for ln in s[i:j].split('\n')[1:]:
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
    else:
        t = ln.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            cur[1].append(t.split(';')[0].strip())
print(name)
for lab, ins in blocks:
    c = lambda p: sum(1 for x in ins if re.match(p, x))
    if len(ins) >= minn:
        tgt = [x.split()[-1] for x in ins if x.startswith('s_cbranch') or x.startswith('s_branch')]
        print('%-10s n %4d mfma %3d valu %4d salu %3d ds %3d vmem %3d wait %3d bar %d -> %s' % (
            lab, len(ins), c(r'v_mfma'), c(r'v_(?!mfma)'), c(r's_(?!waitcnt|barrier|nop|cbranch|branch)'), c(r'ds_'),
            c(r'buffer_|global_|scratch_'), c(r's_waitcnt'), c(r's_barrier'), ','.join(tgt)))
    if dump and lab == dump:
        print('\n'.join(ins))
