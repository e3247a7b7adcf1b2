"""Instruction histogram of one kernel in a -save-temps .s file (tools only)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
m = re.search(r'^(' + pat + r'[A-Za-z0-9_]*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
name, body = m.group(1), m.group(2)
ins = [l.strip().split()[0] for l in body.split('\n')
       if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().endswith(':')]
c = collections.Counter(ins)
print(name, "total instrs", sum(c.values()))
g = collections.Counter()
for k, v in c.items():
    if k.startswith('v_pk'): g['v_pk_*'] += v
    elif k.startswith('v_'): g['v_other'] += v
    elif k.startswith(('ds_', 'global_', 'flat_', 'buffer_', 'scratch_')): g[k] += v
    elif k.startswith('s_waitcnt'): g['s_waitcnt'] += v
    elif k.startswith('s_'): g['s_other'] += v
    else: g[k] += v
for k, v in g.most_common(): print(f"  {k:28s}{v}")
print(c.most_common(24))
md = s[s.index('amdhsa.kernels'):]
blk = md[md.index(name):]
for key in ('.vgpr_count', '.sgpr_count', '.vgpr_spill_count', '.private_segment_fixed_size'):
    mm = re.search(re.escape(key) + r':\s*(\d+)', blk)
    print(key, mm.group(1) if mm else '?')
