"""import alias: the package directory is ``ray-optics_amd/`` (not a valid
Python identifier), so ``import rayoptics_amd`` loads it from there."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'ray-optics_amd')
_spec = _u.spec_from_file_location('rayoptics_amd', _os.path.join(_dir, '__init__.py'),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules['rayoptics_amd'] = _mod
_spec.loader.exec_module(_mod)
