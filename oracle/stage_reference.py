"""TEST INFRASTRUCTURE ONLY -- the recipe that BUILDS the reference into ``oracle/_ref/``.

The reference (mjhoptics/ray-optics) is pure Python, so "building" it means what building a
C reference into a ``.so`` means: the sources stay where they lie under ``/root/reference``,
and only compiled output goes to ``oracle/_ref/`` -- *sourceless byte code* (``name.pyc`` where
``name.py`` would be; CPython imports it when no source is present) plus the reference's
non-code data files (prescriptions ``.roa/.seq/.zmx``, the CODE V three-letter table, the
matplotlib styles) that its importers and its own benchmark read at run time.  No ``.py`` file of
the reference is copied.  ``oracle/_ref/`` is git-ignored (it never enters history) but not
gpurun-ignored: like the built ``.so`` files it travels with the snapshot to the GPU box, where
``/root/reference`` does not exist.

What it is for (and nothing else):
  * ``-m gpu`` tests that run the HIP engine and the *live* reference in one process
    (tests/test_gpu_live_reference.py),
  * ``bench.py``'s ``cpu_baseline`` leg: the reference's own ``rt.trace`` / ``trace.trace_grid``
    timed on the GPU host (``kind: "reference"``).
Nothing under ``ray-optics_amd/`` may import it (tests/test_abi.py enforces that).

The byte code is tied to the interpreter's magic number: the build container and the GPU box run
the same image (CPython 3.10.12).  ``stamp.json`` records it; ``oracle.refshim`` refuses a staged
tree whose magic number differs from the running interpreter's.

    python oracle/stage_reference.py            # (re)build when stale
    python oracle/stage_reference.py --force
"""
import hashlib
import importlib.util
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get('ROX_REFERENCE_TREE', '/root/reference/src')
DST = os.path.join(HERE, '_ref')
PKG = 'rayoptics'
# non-code files the reference reads at run time; images / design files are not needed
DATA_EXT = {'.roa', '.seq', '.zmx', '.ZMX', '.smx', '.csv', '.txt', '.lis', '.mplstyle', '.json',
            '.len', '.sys'}


def _files():
    top = os.path.join(SRC, PKG)
    for d, dirs, names in os.walk(top):
        dirs[:] = sorted(x for x in dirs if x != '__pycache__')
        for n in sorted(names):
            yield os.path.join(d, n)


def source_hash():
    h = hashlib.sha256()
    for p in _files():
        ext = os.path.splitext(p)[1]
        if ext == '.py' or ext in DATA_EXT:
            h.update(os.path.relpath(p, SRC).encode())
            with open(p, 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def present():
    return os.path.isfile(os.path.join(DST, 'stamp.json'))


def stamp():
    with open(os.path.join(DST, 'stamp.json')) as f:
        return json.load(f)


def stage(force=False, verbose=True):
    """byte-compile the reference into oracle/_ref (no-op without /root/reference or when the
    staged tree is current).  Returns the stamp, or None when there is nothing to stage from."""
    if not os.path.isdir(os.path.join(SRC, PKG)):
        return stamp() if present() else None
    sh = source_hash()
    magic = importlib.util.MAGIC_NUMBER.hex()
    if present() and not force:
        st = stamp()
        if st.get('source_hash') == sh and st.get('magic') == magic:
            return st
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    n_code = n_data = 0
    for p in _files():
        rel = os.path.relpath(p, SRC)
        base, ext = os.path.splitext(rel)
        out = os.path.join(DST, rel)
        if ext == '.py':
            os.makedirs(os.path.dirname(out), exist_ok=True)
            py_compile.compile(p, cfile=os.path.join(DST, base + '.pyc'),
                               dfile='<reference>/' + rel, doraise=True,
                               invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
            n_code += 1
        elif ext in DATA_EXT:
            os.makedirs(os.path.dirname(out), exist_ok=True)
            shutil.copyfile(p, out)
            n_data += 1
    st = {'what': 'sourceless byte code + data files of the reference (test infrastructure)',
          'from': SRC, 'source_hash': sh, 'magic': magic,
          'python': sys.version.split()[0], 'modules': n_code, 'data_files': n_data}
    with open(os.path.join(DST, 'stamp.json'), 'w') as f:
        json.dump(st, f, indent=1)
    if verbose:
        print(f'oracle/_ref: {n_code} modules compiled, {n_data} data files '
              f'(reference {sh}, python {st["python"]})')
    return st


if __name__ == '__main__':
    s = stage(force='--force' in sys.argv)
    print(json.dumps(s))
