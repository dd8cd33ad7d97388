"""BASELINE.json configurations as self-contained data (ray-optics_amd/data/):
surface table + per-field ray-start constants, extracted from the live
reference by tests/golden/make_golden.py so that they run where the reference
is not installed."""
import json
import os

from .table import SurfaceTable, field_struct

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


class Workload:
    def __init__(self, name):
        with open(os.path.join(_DATA, name + '.json')) as f:
            d = json.load(f)
        self.name = name
        self.description = d['description']
        self.table = SurfaceTable.from_dict(d['table'])
        self.foc = d['foc']
        self.ref_wvl_idx = d['ref_wvl_idx']
        self.aim = d.get('aim')     # chief-ray aiming problems + the reference's answers
        self.fields = []
        self.image_pts = []
        for fd in d['fields']:
            a = fd['field']
            self.fields.append(field_struct(a[0:3], a[3:5], a[5], a[6], a[7:11], a[11]))
            self.image_pts.append(tuple(fd['image_pt']))

    @property
    def n_ifcs(self):
        return self.table.n_ifcs


def load(name):
    return Workload(name)


class TableField:
    """the slice of the reference's ``Field`` (rayoptics/raytr/opticalspec.py:1119-1353)
    the trace drop-ins read, backed by prebuilt ``rox_field`` constants"""

    def __init__(self, rox_field, image_pt):
        self.rox_field = rox_field
        self.vlx, self.vux = rox_field.vlx, rox_field.vux
        self.vly, self.vuy = rox_field.vly, rox_field.vuy
        self.aim_info = (rox_field.aim[0], rox_field.aim[1])
        self.chief_ray = None
        self.ref_sphere = (image_pt, None, None, None)


class _TableSeq:
    def __init__(self, wl):
        self.surface_table = wl.table           # session.engine_for uses it as is
        self.z_dir = [r.z_dir for r in wl.table.rows]
        self.ifcs = [None] * wl.table.n_ifcs
        self._wl = wl

    def central_wavelength(self):
        return self._wl.table.wvls[self._wl.ref_wvl_idx]


class TableModel:
    """A table-backed stand-in for the reference's ``OpticalModel`` where the
    reference is not installed (the GPU box): ``model['seq_model']`` carries the
    prebuilt :class:`~.table.SurfaceTable`, ``model.fields[i]`` the prebuilt ray
    start constants.  The product entry points (``trace.trace_grid_spot``,
    ``analyses.trace_rays_soa`` ...) run on it unchanged; what it skips is the
    *extraction* of table and field constants from a live model."""

    def __init__(self, name_or_workload):
        wl = load(name_or_workload) if isinstance(name_or_workload, str) else name_or_workload
        self.workload = wl
        self.seq_model = _TableSeq(wl)
        self.fields = [TableField(f, ip) for f, ip in zip(wl.fields, wl.image_pts)]
        self.foc = wl.foc

    def __getitem__(self, key):
        if key in ('seq_model', 'sm'):
            return self.seq_model
        raise KeyError(f'{key!r}: a TableModel carries a surface table and field constants only')
