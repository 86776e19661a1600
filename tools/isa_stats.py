"""Instruction-mix summary of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):', s, re.M):
    name = m.group(1)
    body = s[m.start():]
    end = body.find('.Lfunc_end')
    if end < 0: continue
    body = body[:end]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter(l.split()[0] for l in lines if not l.endswith(':'))
    groups = ['v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_div', 'v_rcp_f64', 'v_rsq_f64', 'v_sqrt', 'v_ldexp', 'v_trig', 'v_fract',
              'ds_read', 'ds_write', 'ds_bpermute', 'global_load', 'global_store', 'scratch_', 's_load', 's_waitcnt',
              's_barrier', 'v_cndmask', 'v_cmp', 'v_mov', 'v_accvgpr', 's_cbranch', 'v_readlane', 'v_readfirstlane']
    g = {k: sum(v for kk, v in c.items() if kk.startswith(k)) for k in groups}
    vg = re.search(r'\.vgpr_count:\s+(\d+)', s[m.start():])
    print(name[:70], 'total', sum(c.values()), {k: v for k, v in g.items() if v})
