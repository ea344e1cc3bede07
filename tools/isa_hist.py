"""Instruction-mix histogram of one kernel in a hipcc -save-temps .s file (static counts, loops not weighted).
usage: python tools/isa_hist.py file.s mangled_substring [start_line end_line]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.split(';')[0].rstrip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.section') or lines[i].strip() == 's_endpgm')
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (start, end)
h = collections.Counter()
for l in lines[lo:hi]:
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    op = l.split()[0]
    cls = ('fma/mul/add f32' if re.match(r'v_(fma|fmac|mul|add|sub|mac)_f32', op) else
           'pk f32' if op.startswith('v_pk_') else
           'lds read' if op.startswith('ds_read') or op.startswith('ds_load') else
           'lds write' if op.startswith('ds_write') or op.startswith('ds_store') else
           'dpp/perm/readlane' if 'dpp' in l or op.startswith(('v_readlane', 'v_readfirstlane', 'ds_bpermute', 'ds_swizzle', 'v_permlane')) else
           'global/scratch' if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')) else
           'cndmask/cmp' if op.startswith(('v_cndmask', 'v_cmp')) else
           'v_mov' if op.startswith(('v_mov', 'v_accvgpr')) else
           'int valu' if re.match(r'v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|min|max|add3|lshl_add|mad_u)', op) and 'f32' not in op else
           'other valu' if op.startswith('v_') else
           'waitcnt' if op.startswith('s_waitcnt') else
           'barrier' if op.startswith('s_barrier') else
           'salu/branch')
    h[cls] += 1
tot = sum(h.values())
print(f'{key}: lines {lo}-{hi}, {tot} instructions')
for k, v in h.most_common():
    print(f'  {k:22s} {v:6d} {100*v/tot:5.1f}%')
