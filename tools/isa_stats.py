"""instruction mix of kernels in a hipcc -S listing: isa_stats.py file.s name-fragment [name-fragment ...]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
starts = [(i, re.match(r'^(_Z\w+):', l).group(1)) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for idx, (i, name) in enumerate(starts):
    if not any(t in name for t in sys.argv[2:]):
        continue
    ins = []
    for l in lines[i + 1:]:
        l = l.split(';')[0].strip()
        if not l or l.startswith(('.', '//')) or l.endswith(':'):
            continue
        ins.append(l.split()[0])
        if l.startswith('s_endpgm'):
            break
    c = collections.Counter(ins)
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith('v_') and 'f64' in k: g['v_f64'] += v
        elif 'lane' in k: g['lane'] += v
        elif k.startswith('v_'): g['v_other'] += v
        elif k.startswith('s_'): g['salu'] += v
        elif k.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): g['vmem'] += v
        elif k.startswith('ds_'): g['lds'] += v
        else: g['other'] += v
    print(name[:60], len(ins), dict(g))
    print('   ', c.most_common(22))
