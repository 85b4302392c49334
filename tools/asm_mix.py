"""Static instruction mix of one kernel in a gfx950 .s file: python tools/asm_mix.py build/asm/sceneflow.s k_scene_flow_v4"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r'^_Z\S*' + sys.argv[2] + r'\S*:', l):
        start = i
    if start is not None and 's_endpgm' in l and i > start:
        end = i
        break
c = collections.Counter()
for line in lines[start + 1:end]:
    line = line.strip()
    if not line or line.startswith(('.', ';')) or line.endswith(':'):
        continue
    c[line.split()[0]] += 1
g = collections.Counter()
for op, n in c.items():
    if op.startswith('v_') and 'f64' in op: g['valu_f64'] += n
    elif op.startswith('v_'): g['valu_other'] += n
    elif op.startswith('s_'): g['salu'] += n
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): g['vmem'] += n
    elif op.startswith('ds_'): g['lds'] += n
    else: g[op] += n
print(sum(c.values()), dict(g))
print(sorted(((n, op) for op, n in c.items()), reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 30])
