"""builds libroxtrace.so (HIP kernels + C ABI) in-tree for gfx950.

    python ray-optics_amd/build.py [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: the
kernels restate NumPy's unfused elementwise arithmetic and spell the BLAS dot
sites as explicit fma() (see csrc/roxtrace.hip header)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'roxtrace.hip')
HDR = os.path.join(HERE, '..', 'include', 'roxtrace.h')
LIB = os.path.join(HERE, 'libroxtrace.so')

FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-ffp-contract=off', '-fno-fast-math', '-Wall', '-Wno-unused-function']


def hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in (SRC, HDR, __file__))


def build(force=False, extra=(), out=None):
    """out != None builds an experiment variant (tools/ab_bench.py) next to the
    product library without touching it"""
    lib = out or LIB
    if out is None and not force and not stale():
        return LIB
    cmd = [hipcc(), *FLAGS, *extra, '-o', lib, SRC]
    subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    argv = sys.argv[1:]
    out = None
    if '--out' in argv:
        i = argv.index('--out')
        out = argv[i + 1]
        del argv[i:i + 2]
    print(build(force='--force' in argv, out=out,
                extra=[a for a in argv if a != '--force']))
