"""usage: python tools/isa_count.py <mangled kernel name prefix> ...   -- static instruction counts of kernels in wfsim_amd/csrc/wfs_engine.s (make asm)"""
import sys, os, re
txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'wfsim_amd', 'csrc', 'wfs_engine.s')).read()
for name in sys.argv[1:]:
    i = txt.index('\n' + name); j = txt.index('.end_amdhsa_kernel', i)
    ops = []
    for l in txt[i:j].split('\n'):
        t = l.split()
        if l.startswith('\t') and t and not t[0].startswith(('.', ';')):
            ops.append(t[0])
    c = lambda f: sum(1 for o in ops if f(o))
    print(name, dict(valu=c(lambda o: o.startswith('v_')), salu=c(lambda o: o.startswith('s_')), ds=c(lambda o: o.startswith('ds_')), vmem=c(lambda o: o.startswith(('global_', 'buffer_', 'flat_'))),
                     fma=c(lambda o: 'fma_f64' in o), mul64=c(lambda o: 'mul_f64' in o), add64=c(lambda o: 'add_f64' in o), mad_u64=c(lambda o: 'mad_u64' in o),
                     waitcnt=c(lambda o: o == 's_waitcnt'), barrier=c(lambda o: o == 's_barrier'), saveexec=c(lambda o: 'saveexec' in o), total=len(ops)))
