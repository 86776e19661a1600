"""Register / scratch / LDS metadata of every kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):
tools/kernel_meta.py file.s [name-filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = s[s.index('amdhsa.kernels'):]
for blk in meta.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(aslr::KArgs.*', '', dn).replace('void aslr::', '')
    if flt and flt not in dn: continue
    print('%-60s agpr %3s vgpr %3s sgpr %3s scratch %5s lds %6s spilled v %4s s %4s' % (
        dn[:60], blk.split()[0], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'),
        g('group_segment_fixed_size'), g('vgpr_spill_count'), g('sgpr_spill_count')))
