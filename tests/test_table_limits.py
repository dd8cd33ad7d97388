"""What SurfaceTable refuses (UnsupportedModelError), on stand-in interface objects carrying the
attributes the reference's classes carry: nothing outside the kernels' scope is traced silently."""
import numpy as np
import pytest

from rayoptics_amd import SurfaceTable, UnsupportedModelError, abi


class Spherical:
    def __init__(self, cv=0.0):
        self.cv = cv


class EvenPolynomial:
    def __init__(self, cv, coefs):
        self.cv, self.cc, self.ec, self.coefs = cv, 0.0, 1.0, list(coefs)


class Circular:
    def __init__(self, radius=1.0, **kw):
        self.radius = radius
        self.__dict__.update(kw)


class Rectangular:
    def __init__(self, x, y):
        self.x_half_width, self.y_half_width = x, y


class Elliptical(Rectangular):
    pass


class Hexagonal:
    pass


class Ifc:
    def __init__(self, profile, mode='transmit', ca=None, **kw):
        self.profile, self.interact_mode, self.max_aperture = profile, mode, 1.0
        self.clear_apertures = ca or []
        self.__dict__.update(kw)


def path_of(ifc):
    eye = np.identity(3)
    return [(Ifc(Spherical(), 'dummy'), None, (eye, np.array([0., 0., 10.])), 1.0, 1.0),
            (ifc, None, (eye, np.array([0., 0., 5.])), 1.5, 1.0),
            (Ifc(Spherical(), 'dummy'), None, None, 1.0, 1.0)]


def test_limits_and_unknown_classes_raise():
    ok = SurfaceTable.from_paths([path_of(Ifc(EvenPolynomial(0.01, [0.0] * 12)))], [550.0])
    assert ok.rows[1].ncoef == 0                            # twelve zeros: max_nonzero_coef = 0
    ok = SurfaceTable.from_paths([path_of(Ifc(EvenPolynomial(0.01, [1e-4, 0, 1e-9])))], [550.0])
    assert ok.rows[1].ncoef == 3 and ok.rows[1].profile == abi.PROFILE_NAMES['EvenPolynomial']
    with pytest.raises(UnsupportedModelError):              # 11 live coefficients
        SurfaceTable.from_paths([path_of(Ifc(EvenPolynomial(0.01, [1e-9] * 11)))], [550.0])
    with pytest.raises(UnsupportedModelError):              # five clear apertures
        SurfaceTable.from_paths([path_of(Ifc(Spherical(0.02), ca=[Circular()] * 5))], [550.0])
    with pytest.raises(UnsupportedModelError):              # an aperture class the kernels do not know
        SurfaceTable.from_paths([path_of(Ifc(Spherical(0.02), ca=[Hexagonal()]))], [550.0])
    with pytest.raises(UnsupportedModelError):              # a profile class they do not know

        class Biconic(Spherical):
            pass
        SurfaceTable.from_paths([path_of(Ifc(Biconic(0.02)))], [550.0])
    with pytest.raises(UnsupportedModelError):              # raytrace.py:205 would fail on None too
        SurfaceTable.from_paths([path_of(Ifc(Spherical(0.02), phase_element=None))], [550.0])

    class NoProfile:
        interact_mode, max_aperture = 'transmit', 1.0
    with pytest.raises(UnsupportedModelError):
        SurfaceTable.from_paths([path_of(NoProfile())], [550.0])


def test_aperture_rows():
    t = SurfaceTable.from_paths([path_of(Ifc(Spherical(0.02), ca=[
        Circular(3.0, x_offset=0.5, is_obscuration=True), Rectangular(2.0, 1.0), Elliptical(2.0, 1.0)]))],
        [550.0])
    r = t.rows[1]
    assert r.n_ap == 3
    assert (r.ap[0].kind, r.ap[0].a, r.ap[0].x_offset, r.ap[0].is_obscuration) == (abi.AP_CIRCULAR, 3.0, 0.5, 1)
    assert (r.ap[1].kind, r.ap[1].a, r.ap[1].b) == (abi.AP_RECTANGULAR, 2.0, 1.0)
    assert r.ap[2].kind == abi.AP_ALWAYS_BLOCK              # Elliptical has no point_inside: blocks


def test_from_prescription_rows():
    t = SurfaceTable.from_prescription([
        dict(cv=0, thi=1e10), dict(cv=0.02, thi=3.0, n=[1.52, 1.51], profile='Conic', cc=-0.5),
        dict(cv=-0.01, thi=40.0, mode='reflect'),
        dict(profile='RadialPolynomial', cv=0.03, ec=0.7, coefs=[0, 1e-5, 0], thi=2.0, max_aperture=4.0),
        dict(cv=0, thi=0)], wvls=[500.0, 600.0])
    assert t.n_ifcs == 5 and t.n_table.shape == (2, 5)
    assert t.rows[1].cc == -0.5 and t.rows[1].ec == 0.5 and t.n_table[1, 1] == 1.51
    assert t.rows[2].mode == abi.MODE_NAMES['reflect'] and t.rows[2].z_dir == -1.0 and t.rows[3].z_dir == -1.0
    assert t.rows[3].ec == 0.7 and abs(t.rows[3].cc + 0.3) < 1e-15 and t.rows[3].ncoef == 2
    assert t.rows[3].max_aperture == 4.0
    with pytest.raises(UnsupportedModelError):
        SurfaceTable.from_prescription([dict(cv=0, thi=1), dict(profile='EvenPolynomial', coefs=[1e-9] * 11, thi=1),
                                        dict(cv=0)])
    back = SurfaceTable.from_dict(t.to_dict())
    assert bytes(back.rows) == bytes(t.rows) and np.array_equal(back.n_table, t.n_table)
