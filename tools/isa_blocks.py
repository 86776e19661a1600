"""Basic blocks of one kernel in a gfx950 assembly file (hipcc -S --cuda-device-only): label, instruction count by
class, terminator -- to see which static code a loop iteration executes.  Usage: isa_blocks.py file.s KERNEL_SUBSTRING"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = [x for x in re.finditer(r'^(_Z\w+):', s, re.M) if key in x.group(1)][0]
body = s[m.start():]
body = body[:body.find('.Lfunc_end')]
blocks, cur, name = [], [], "entry"
for l in body.split('\n')[1:]:
    t = l.strip()
    if not t or t.startswith(';'): continue
    if re.match(r'^\.LBB[\w]+:', t):
        blocks.append((name, cur)); cur, name = [], t.split(':')[0]; continue
    if t.startswith('.'): continue
    cur.append(t.split(';')[0].strip())
blocks.append((name, cur))
def cls(op):
    if op.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64', 'v_rsq_f64', 'v_rcp_f64', 'v_div', 'v_max_f64', 'v_min_f64', 'v_ldexp_f64', 'v_fract', 'v_trig', 'v_sqrt')): return 'f64'
    if op.startswith('v_accvgpr'): return 'agpr'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith('v_mov'): return 'mov'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'
tot = Counter()
for name, ins in blocks:
    c = Counter(cls(i.split()[0]) for i in ins)
    tot.update(c)
    term = [i for i in ins if i.startswith(('s_cbranch', 's_branch', 's_endpgm'))]
    print("%-12s %4d  %s  -> %s" % (name, len(ins), dict(c), ' | '.join(t for t in term[-2:])))
print("total", sum(tot.values()), dict(tot))
