"""packaging of the plugin INTEGRATION.md describes: the package lives in ``ray-optics_amd/``
(the directory name the project layout fixes; not a valid identifier) and installs under its
import name ``rayoptics_amd``, with the in-tree HIP library, the kernel sources it is rebuilt
from when stale, and the stored workloads.

    python ray-optics_amd/build.py          # libroxtrace.so for gfx950 (hipcc)
    pip install -e .                        # or: pip install .
"""
from setuptools import setup

setup(
    name='rayoptics-amd',
    version='0.4.0',
    description="MI355X (gfx950) engine for ray-optics' sequential real-ray trace hot path: "
                'hand-written HIP kernels behind a C ABI, drop-ins for rayoptics.raytr',
    python_requires='>=3.10',
    install_requires=['numpy>=1.24'],
    extras_require={'gpu': ['torch'], 'rayoptics': ['rayoptics']},
    packages=['rayoptics_amd'],
    package_dir={'rayoptics_amd': 'ray-optics_amd'},
    package_data={'rayoptics_amd': ['libroxtrace.so', 'libroxtrace.so.srchash', 'csrc/*', 'data/*.json']},
    include_package_data=False,
)
