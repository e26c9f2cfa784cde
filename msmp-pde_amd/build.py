"""Build libmsmp_pde.so (HIP, gfx950 only) in-tree with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/msmp_pde.h); the Python host binds it with ctypes."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
VARIANT = os.environ.get('MSMP_TILE_VARIANT', '')      # kernel A/B experiments (-DMSMP_TILE_VARIANT=N): separate library libmsmp_pde_v<N>.so
PRECISE = os.environ.get('MSMP_PRECISE', '') == '1'     # diagnostic build: libm / correctly rounded activations (msmp_common.h)
LOLO = os.environ.get('MSMP_LOLO', '')                 # diagnostic build: fourth product lo*lo of the fp16 split (mfma_tiles.h), level 1 | 2 -> libmsmp_pde_lolo<N>.so
LIB = os.path.join(PKG, f'libmsmp_pde_lolo{LOLO}.so' if LOLO else 'libmsmp_pde_precise.so' if PRECISE else ('libmsmp_pde_prof.so' if os.environ.get('MSMP_PROF') else (f'libmsmp_pde_v{VARIANT}.so' if VARIANT else 'libmsmp_pde.so')))

SOURCES = {  # file -> extra flags
    'mlp_kernels.hip': [],
    'tile_kernels.hip': [],
    'aux_kernels.hip': [],
    # MFMA results stay in VGPRs (the activations read them with VALU; the stationary weights take the AGPRs)
    # -fno-slp-vectorize: the anti-phased kernel's state update stays UNPACKED fp32 (a packed-fp32 instruction beside the partner
    # wave's MFMAs costs more cycles than the two instructions it replaces: 2 480 vs 1 960 cycles per vector half, scripts/prof_lem.py;
    # the SLP vectoriser would re-pack the scalar ops)
    'lem_kernel.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form', '-fno-slp-vectorize'],
    'lem_train_kernel.hip': [],
    'train_kernels.hip': [],
    'mlp2_kernel.hip': [],
    'wide_kernels.hip': [],
    'decoder_kernel.hip': [],
    'graph_kernels.hip': ['-ffp-contract=off'],   # float64 distance compares must round like the host's
}
PROF = {'gw': ['-DMSMP_PROF_GW=1'], 'lem': ['-DMSMP_PROF_LEM=1'], 'tile': ['-DMSMP_PROF_TILE=1'], '1': ['-DMSMP_PROF=1'], 'edge': ['-DMSMP_PROF=1', '-DMSMP_PROF_EDGE=1'], 'proj': ['-DMSMP_PROF=1', '-DMSMP_PROF_PROJ=1']}.get(os.environ.get('MSMP_PROF', ''), [])     # phase counters in the tail kernel (scripts/prof_tail.py)
COMMON = PROF + ([f'-DMSMP_LOLO={LOLO}'] if LOLO else []) + ([f'-DMSMP_TILE_VARIANT={VARIANT}'] if VARIANT else []) + (['-DMSMP_PRECISE_ACT=1'] if PRECISE else []) + ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
          '-fvisibility=hidden', '-fvisibility-inlines-hidden',
          '-I', os.path.join(ROOT, 'include'), '-I', CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(ROOT, 'include', 'msmp_pde.h'))
    headers.append(os.path.abspath(__file__))          # the flags live here
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', f'.lolo{LOLO}.o' if LOLO else '.precise.o' if PRECISE else ('.prof.o' if os.environ.get('MSMP_PROF') else (f'.v{VARIANT}.o' if VARIANT else '.o'))))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        # -Bsymbolic: the library's own references (rocPRIM templates, ...) bind to its own definitions
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,-Bsymbolic', '-Wl,--exclude-libs,ALL', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
