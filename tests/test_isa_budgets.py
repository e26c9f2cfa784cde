"""Static checks on the gfx950 code of the hot kernels (no GPU: hipcc cross-compiles here).  They pin what round 4 found by reading the ISA
 and what a later edit can silently undo:
  * register budgets that decide the resident workgroups per CU (a volatile guard asm took the gather message kernel from 128 to 194);
  * no scratch in the message / LEM kernels;
  * an LDS-DMA prefetch is not waited for where it is issued (a second __shared__ array in the node tail brings that back)."""
import os, re, shutil, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'msmp-pde_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not found')


@pytest.fixture(scope='module')
def isa(tmp_path_factory):
    out = {}
    d = tmp_path_factory.mktemp('isa')
    for src in ('mlp_kernels.hip', 'tile_kernels.hip', 'lem_kernel.hip'):
        s = d / (src + '.s')
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-S', '--cuda-device-only',
                        '-o', str(s), os.path.join(CSRC, src)], check=True, capture_output=True, cwd=str(d))
        out[src] = open(s).read()
    return out


def kernel(text, prefix):
    m = re.search(r'^(' + re.escape(prefix) + r'\w*):[^\n]*\n(.*?)s_endpgm(.*?)\.end_amdhsa_kernel', text, re.S | re.M)
    assert m, prefix
    body, meta = m.group(2), m.group(3)
    num = lambda key: int(re.search(r'\.amdhsa_' + key + r'\s+(\d+)', meta).group(1))
    return body, num


@pytest.mark.parametrize('src,prefix,max_vgpr', [
    ('tile_kernels.hip', '_ZN4msmp16edge_tile_kernelILi2ELb0E', 168),        # three workgroups per CU
    ('tile_kernels.hip', '_ZN4msmp16edge_tile_kernelILi2ELb1E', 168),
    ('mlp_kernels.hip', '_ZN4msmp20edge_mlp_kernel_occ2ILi1ELb1ELb1ELb1E', 128),   # the gather message kernel (RPU): four waves per SIMD
    ('mlp_kernels.hip', '_ZN4msmp22node_tail_split_kernelILb1E', 256),
    ('lem_kernel.hip', '_ZN4msmp22lem_encoder_ws3_kernelILi4ELi1E', 256),
    ('lem_kernel.hip', '_ZN4msmp22lem_encoder_ws3_kernelILi6ELi2E', 256),
])
def test_register_budgets(isa, src, prefix, max_vgpr):
    _, num = kernel(isa[src], prefix)
    vgpr = num('next_free_vgpr')
    assert vgpr <= max_vgpr, f'{prefix}: {vgpr} vector registers (budget {max_vgpr})'
    # (the gated tail and the 2-D LEM instantiations live at the 256-register limit and keep two or three values in scratch, outside their loops)
    budget = 24 if 'node_tail' in prefix or 'ELi2E' in prefix.split('ws3_kernel')[-1] else 0
    assert num('private_segment_fixed_size') <= budget, f'{prefix}: {num("private_segment_fixed_size")} bytes of scratch (budget {budget})'


def wait_distances(body):
    """(distance in instructions, barriers between) of every vmcnt wait to the youngest load it completes"""
    ins = [t.strip() for t in body.splitlines() if t.strip() and not t.strip().startswith((';', '.'))]
    pend, nb, out = [], 0, []
    for k, t in enumerate(ins):
        if t.startswith('s_barrier'): nb += 1
        if re.match(r'(global|buffer)_load', t): pend.append((k, nb, 'lds' in t.split()[0]))
        m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
        if m:
            done = pend[:max(len(pend) - int(m.group(1)), 0)]
            if done: out.append((k - done[-1][0], nb - done[-1][1], done[-1][2], k))
            pend = pend[len(done):]
    return out


def test_node_tail_does_not_wait_for_its_lds_dma_prefetch(isa):
    body, _ = kernel(isa['mlp_kernels.hip'], '_ZN4msmp22node_tail_split_kernelILb1E')
    early = [w for w in wait_distances(body) if w[2] and w[0] < 40 and w[3] > 400]      # (past the kernel's prologue)
    # one is expected: the second head's prologue (its first weight chunk and its bias batch are both needed at once)
    assert len(early) <= 1, f'LDS-DMA waited for {early[0][0]} instructions behind its request (instr {early[0][3]}): see DESIGN 4.18 (one __shared__ object; requests pinned)'


def test_message_kernel_k_loop_waits_for_its_dma_at_the_barrier(isa):
    body, _ = kernel(isa['tile_kernels.hip'], '_ZN4msmp16edge_tile_kernelILi2ELb0E')
    dma = [w for w in wait_distances(body) if w[2]]
    assert dma and all(w[0] >= 60 for w in dma), dma


def _regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'v\[(\d+):(\d+)\]$', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.parametrize('src', ['mlp_kernels.hip', 'tile_kernels.hip', 'lem_kernel.hip'])
def test_no_vector_write_in_the_slot_in_front_of_an_mfma_that_reads_it(isa, src):
    """gfx950 does not interlock a VALU write against an MFMA reading the register as SrcA / SrcB in the next issue slot (DESIGN.md 4.18,
    profiles/r04N_mfma_operand_hazard.md).  The compiler pads its own instructions; inline asm is guarded in the source (mfma_operand_guard).
    This checks the result: in no kernel of the file does a vector instruction write an operand of the MFMA right behind it."""
    prev, bad = None, []
    for line in isa[src].splitlines():
        t = line.strip()
        if not t or t.startswith((';', '.')) or t.endswith(':'): continue
        if t.startswith('v_mfma') and prev and prev.startswith('v_') and not prev.startswith(('v_mfma', 'v_cmp', 'v_readlane', 'v_readfirstlane')):
            ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
            written = _regs(prev.split(None, 1)[1].split(',')[0])
            if written & (_regs(ops[1]) | _regs(ops[2])): bad.append((prev, t))
        prev = t
    assert not bad, f'{len(bad)} unguarded VALU -> MFMA operand pairs, e.g. {bad[0]}'
