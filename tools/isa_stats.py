"""Per-kernel ISA summary of a hipcc -save-temps .s file: registers, spills, scratch, flat/scratch memory ops, v_perm,
vmcnt(0) waits -- the things that went wrong silently in the wave-specialised kernels ("compiler pitfalls")."""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
meta = {}
for m in re.finditer(r'- \.agpr_count:\s*(\d+)(.*?)\.wavefront_size', s, re.S):
    blk = m.group(0)
    name = re.search(r'\.name:\s*(\S+)', blk).group(1)
    g = lambda k: int(re.search(r'\.%s:\s*(\d+)' % k, blk).group(1))
    meta[name] = dict(agpr=g('agpr_count'), vgpr=g('vgpr_count'), sgpr=g('sgpr_count'), spill=g('vgpr_spill_count'),
                      scratch=g('private_segment_fixed_size'), lds=g('group_segment_fixed_size'))
for name, md in meta.items():
    if pat and pat not in name:
        continue
    i = re.search(r'^%s:' % re.escape(name), s, re.M).start()
    j = s.find('.end_amdhsa_kernel', i)
    code = s[i:j]
    c = lambda p: len(re.findall(p, code))
    try:
        dem = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'\(anonymous namespace\)::', '', dem).split('(')[0].replace('void ', '')
    except Exception:
        dem = name
    print('%-50s v%3d a%3d spill %3d scratch %4d | instr %5d mfma %4d flat %3d scr %3d perm %3d vmcnt0 %3d vmcnt %3d' % (
        dem[-50:], md['vgpr'], md['agpr'], md['spill'], md['scratch'], code.count('\n'), c(r'v_mfma'), c(r'\bflat_'),
        c(r'\bscratch_'), c('v_perm_b32'), c(r'vmcnt\(0\)'), c('vmcnt')))
