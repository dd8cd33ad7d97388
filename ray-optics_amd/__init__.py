"""rayoptics_amd -- MI355X (gfx950) engine for ray-optics' sequential real-ray
trace hot path (rayoptics.raytr.raytrace.trace/trace_raw and the grid / fan /
list drivers of rayoptics.raytr.trace and rayoptics.raytr.analyses).

The directory is named ``ray-optics_amd``; import it as ``rayoptics_amd``
(``rayoptics_amd.py`` at the repository root is the alias loader).

Layout:
  csrc/        hand-written HIP kernels + the C ABI (libroxtrace.so)
  abi.py       ctypes mirror of include/roxtrace.h
  table.py     SequentialModel -> flat surface table (read-only extraction)
  engine.py    handle + launch wrappers over the C ABI (torch = device memory)
  raypkg.py    lazy RayPkg/RaySeg views over SoA results
  trace.py     drop-ins for rayoptics.raytr.trace.{trace_grid,trace_fan,...}
  analyses.py  drop-ins for rayoptics.raytr.analyses.{trace_list_of_rays,...}
  dist.py      one-process-per-GPU sharding of (field x wavelength x pupil rows)
  install.py   rebinding of the reference's module-level seams

The product path never imports anything under ``oracle/`` and has no CPU
fallback: without libroxtrace.so every trace entry raises.
"""
from . import abi                                  # noqa: F401
from .table import SurfaceTable, UnsupportedModelError, field_struct  # noqa: F401

__all__ = ['abi', 'SurfaceTable', 'UnsupportedModelError', 'field_struct']
