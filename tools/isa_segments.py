"""usage: python tools/isa_segments.py <mangled kernel name prefix>   -- basic blocks of a kernel in wfs_engine.s with instruction counts"""
import re, sys, os
txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'wfsim_amd', 'csrc', 'wfs_engine.s')).read()
i = txt.index('\n' + sys.argv[1]); j = txt.index('.end_amdhsa_kernel', i)
lines = txt[i:j].split('\n')
seg = []; cur = dict(label='entry', v=0, ds=0, g=0, bar=0, s=0, br=[], f64=0, line=0, wait=0)
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB[0-9_]+):', l)
    if m:
        seg.append(cur); cur = dict(label=m.group(1), v=0, ds=0, g=0, bar=0, s=0, br=[], f64=0, line=n, wait=0)
        continue
    t = l.strip().split()
    if not t or t[0].startswith((';', '.')): continue
    op = t[0]
    if op.startswith('v_'):
        cur['v'] += 1
        if 'f64' in op: cur['f64'] += 1
    elif op.startswith('ds_'): cur['ds'] += 1
    elif op.startswith(('global_', 'buffer_')): cur['g'] += 1
    elif op == 's_barrier': cur['bar'] += 1
    elif op == 's_waitcnt': cur['wait'] += 1
    elif op.startswith('s_cbranch') or op == 's_branch': cur['br'].append(op[2:] + '->' + t[1])
    elif op.startswith('s_'): cur['s'] += 1
seg.append(cur)
for s in seg:
    if s['v'] + s['ds'] + s['g'] + s['bar'] > 0 or s['br']:
        print(f"{s['label']:12s} L{s['line']:5d} v={s['v']:4d} f64={s['f64']:3d} ds={s['ds']:3d} g={s['g']:3d} bar={s['bar']} wait={s['wait']} s={s['s']:3d} {' '.join(s['br'])}")
