"""Static instruction census of one kernel of an assembly listing (hipcc -S --cuda-device-only), per barrier-delimited segment:
    python scripts/isa_census.py file.s [mangled kernel name substring]"""
import collections, sys
lines = open(sys.argv[1]).read().splitlines()
if len(sys.argv) > 2:
    a = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l and (l.split(';')[0].rstrip().endswith(':')))
    b = next(i for i in range(a, len(lines)) if 's_endpgm' in lines[i])
    lines = lines[a:b + 1]
seg, cnt = 0, collections.defaultdict(collections.Counter)
for l in lines:
    l = l.strip()
    if not l or l[0] in ';.' or l.endswith(':'):
        continue
    op = l.split()[0]
    if op == 's_barrier':
        seg += 1
        continue
    c = ('mfma' if op.startswith('v_mfma') else 'trans' if op[:5] in ('v_exp', 'v_rcp', 'v_rsq', 'v_sqr', 'v_log') else 'vpk' if op.startswith('v_pk_') else
         'vcvt' if op.startswith('v_cvt') or 'mix' in op else 'vmov' if op.startswith('v_mov') else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else
         'vmem' if op.split('_')[0] in ('global', 'buffer', 'scratch', 'flat') else 'wait' if op.startswith('s_waitcnt') else 'branch' if op.startswith('s_cbranch') or op == 's_branch' else 'salu')
    cnt[seg][c] += 1
keys = ['mfma', 'valu', 'vmov', 'vcvt', 'vpk', 'trans', 'lds', 'vmem', 'salu', 'branch', 'wait']
print('seg ' + ' '.join(f'{k:>6s}' for k in keys))
tot = collections.Counter()
for s in sorted(cnt):
    print(f'{s:3d} ' + ' '.join(f'{cnt[s][k]:6d}' for k in keys))
    tot.update(cnt[s])
print('all ' + ' '.join(f'{tot[k]:6d}' for k in keys))
