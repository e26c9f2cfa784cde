"""For every `s_waitcnt vmcnt(n)` of one kernel in a gfx950 assembly file: which loads it completes, how many instructions behind the
youngest of them it sits and how many barriers lie between (a prefetch waited for a few instructions behind its request is not one).
    python scripts/isa_wait_distance.py /tmp/k.s <mangled kernel name prefix> [max distance to report, default 40]"""
import re, sys
src, name = sys.argv[1], sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 40
L, on = [], False
for l in open(src):
    if re.match(r'^' + re.escape(name) + r'\w*:', l): on = True
    elif on and 's_endpgm' in l: break
    elif on: L.append(l.rstrip())
ins = [t.strip() for t in L if t.strip() and not t.strip().startswith((';', '.'))]
pend, nb, loop_marks = [], 0, 0
for k, t in enumerate(ins):
    if t.startswith('s_barrier'): nb += 1
    if re.match(r'(global|buffer|scratch)_load', t): pend.append((k, nb, ' '.join(t.split()[:2])))
    m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
    if m:
        n = int(m.group(1)); done = pend[:max(len(pend) - n, 0)]
        if done and k - done[-1][0] <= lim:
            print(f'instr {k}: vmcnt({n}) completes {len(done)} load(s); youngest requested {k - done[-1][0]} instructions earlier, {nb - done[-1][1]} barriers between: {done[-1][2]}')
        pend = pend[len(done):]
