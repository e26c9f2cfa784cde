"""Static scan of a gfx950 assembly file for prefetches that are waited for where they are issued: an `s_waitcnt vmcnt(n)` within WINDOW
instructions behind a global / buffer load whose result it (by its count) waits for, inside a loop.
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/k.s csrc/file.hip && python scripts/isa_early_waits.py /tmp/k.s"""
import re, sys
WINDOW = 12
kern = None; body = []
def scan(name, L):
    # loop ranges: from a label to a backward branch to it
    labels = {m.group(1): i for i, l in enumerate(L) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
    loops = []
    for i, l in enumerate(L):
        m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t < i: loops.append((t, i))
    inloop = lambda i: any(a <= i <= b for a, b in loops)
    ins = [(i, l.strip()) for i, l in enumerate(L) if l.strip() and not l.strip().startswith((';', '.'))]
    hits = 0
    for k, (i, t) in enumerate(ins):
        m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
        if not m or not inloop(i): continue
        n = int(m.group(1))
        # loads among the previous WINDOW instructions, youngest first
        young = [j for j in range(k - 1, max(k - 1 - WINDOW, -1), -1) if re.match(r'(global|buffer)_load', ins[j][1]) and 'lds' not in ins[j][1]]
        if len(young) > n: hits += 1
    if hits: print(f'{name[:90]:90s} loops {len(loops):3d}  early waits in loops {hits}')
for l in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        if kern: scan(kern, body)
        kern, body = m.group(1), []
    elif kern: body.append(l.rstrip('\n'))
if kern: scan(kern, body)
