"""builds libroxtrace.so (HIP kernels + C ABI) in-tree for gfx950.

    python ray-optics_amd/build.py [--force] [--out lib.so] [extra hipcc flags]

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: the
kernels restate NumPy's unfused elementwise arithmetic and spell the BLAS dot
sites as explicit fma() (see csrc/rox_device.hpp header).

The trace kernel is compiled once per feature instance (csrc/inst_*.hip), each
in its own translation unit so that the instances build in parallel; objects
are cached under build/obj and rebuilt when a source, a header or the flags
change."""
import concurrent.futures
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INC = os.path.join(HERE, '..', 'include')
LIB = os.path.join(HERE, 'libroxtrace.so')
OBJ_ROOT = os.path.join(HERE, '..', 'build', 'obj')

FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-ffp-contract=off', '-fno-fast-math', '-Wall', '-Wno-unused-function',
         # the dynamic symbol table is the C ABI and nothing else: the entry points get default
         # visibility from the `#pragma GCC visibility push(default)` of include/roxtrace*.h
         '-fvisibility=hidden', '-fvisibility-inlines-hidden']
# DT_SONAME = libroxtrace.so.<ROX_ABI_VERSION>; build() keeps a link of that name beside the library
# for executables linked with -lroxtrace (examples/spot_diagram.c)
SONAME_FMT = 'libroxtrace.so.%d'


def abi_version():
    import re
    with open(os.path.join(INC, 'roxtrace.h')) as f:
        return int(re.search(r'#define\s+ROX_ABI_VERSION\s+(\d+)', f.read()).group(1))


def ensure_soname_link(lib=None):
    """libroxtrace.so.<abi> -> libroxtrace.so next to the library (idempotent)"""
    lib = lib or LIB
    if os.path.basename(lib) != 'libroxtrace.so' or not os.path.isdir(INC):
        return None
    link = os.path.join(os.path.dirname(lib), SONAME_FMT % abi_version())
    for old in glob.glob(os.path.join(os.path.dirname(lib), 'libroxtrace.so.[0-9]*')):
        if old != link and not old.endswith('.srchash'):
            os.remove(old)                      # the link of an earlier ABI version
    try:
        if os.path.islink(link) and os.readlink(link) == 'libroxtrace.so':
            return link
        if os.path.lexists(link):
            os.remove(link)
        os.symlink('libroxtrace.so', link)
    except OSError:
        shutil.copyfile(lib, link)
    return link


def hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, '*.hpp')) + glob.glob(os.path.join(INC, '*.h')))


def source_hash(extra=()):
    """digest of every input of the build: sources, headers, flags, this file"""
    h = hashlib.sha256()
    for p in sources() + headers() + [os.path.join(CSRC, 'libroxtrace.map'), os.path.abspath(__file__)]:
        with open(p, 'rb') as f:
            h.update(os.path.basename(p).encode() + b'\0' + f.read())
    h.update(' '.join(FLAGS + list(extra)).encode())
    return h.hexdigest()


def _stamp(lib):
    return lib + '.srchash'


def stale(lib=None, extra=()):
    lib = lib or LIB
    if not os.path.isdir(INC):
        # an installed copy (`pip install .`) without the repository around it: the C ABI
        # headers are not there to rebuild from -- the library that was installed is the one
        return not os.path.exists(lib)
    if not os.path.exists(lib) or not os.path.exists(_stamp(lib)):
        return True
    with open(_stamp(lib)) as f:
        return f.read().strip() != source_hash(extra)


def build(force=False, extra=(), out=None):
    """out != None builds an experiment variant (tools/ab_bench.py) next to the
    product library without touching it"""
    lib = out or LIB
    extra = list(extra)
    if not force and not stale(lib, extra):
        ensure_soname_link(lib)
        return lib
    digest = source_hash(extra)
    objdir = os.path.join(OBJ_ROOT, digest[:16])
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        if not os.path.exists(obj):
            tmp = obj + f'.tmp{os.getpid()}'
            subprocess.check_call([cc, *FLAGS, *extra, '-c', src, '-o', tmp])
            os.replace(tmp, obj)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        objs = list(ex.map(compile_one, sources()))
    tmp = lib + f'.tmp{os.getpid()}'
    subprocess.check_call([cc, '--offload-arch=gfx950', '-shared', '-fPIC',
                           '-Wl,-soname,' + SONAME_FMT % abi_version(),
                           '-Wl,--version-script=' + os.path.join(CSRC, 'libroxtrace.map'),
                           '-o', tmp, *objs])
    os.replace(tmp, lib)
    ensure_soname_link(lib)
    with open(_stamp(lib), 'w') as f:
        f.write(digest + '\n')
    # keep the object cache small (it travels with the tree): the three newest digests
    try:
        dirs = sorted((os.path.join(OBJ_ROOT, d) for d in os.listdir(OBJ_ROOT)),
                      key=os.path.getmtime, reverse=True)
        for d in dirs[3:]:
            shutil.rmtree(d, ignore_errors=True)
    except OSError:
        pass
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
