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
