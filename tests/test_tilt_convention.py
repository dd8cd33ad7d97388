"""The tilt convention, pinned against an independent implementation.

The reference builds a tilted interface's rotation with
``transforms3d.euler.euler2mat(*deg2rad([-alpha, -beta, gamma]), axes='rxyz')``
(rayoptics/util/misc_math.py:150-160).  transforms3d is a third-party package
that is absent here, so the matrix is restated twice: ``oracle/refshim.py``
(the stand-in the live reference imports in the tests) and
``ingest._euler2rot3d`` (decentered prescriptions).  SciPy *is* installed, and
``Rotation.from_euler('XYZ', ...)`` (capital letters = intrinsic rotations about
x, then y, then z) is a separately written implementation of the same
convention that goes through quaternions: agreement to a few ulp on random
angles pins the axis order, the handedness and the sign flips of alpha / beta.
"""
import math

import numpy as np
import pytest

from scipy.spatial.transform import Rotation

import rayoptics_amd  # noqa: F401
from rayoptics_amd import ingest
from oracle import refshim

ULP = np.finfo(float).eps


def _angles(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-180., 180., size=(n, 3))
    # the cases prescriptions actually carry: single-axis tilts, folds, zeros
    special = np.array([[0., 0., 0.], [45., 0., 0.], [0., -30., 0.], [0., 0., 90.], [90., 90., 0.],
                        [-8.5, 0., 0.], [0., 0., -0.0], [180., 0., 0.], [1e-9, -1e-9, 1e-9]])
    return np.vstack([special, a])


def test_ingest_tilt_matrix_matches_scipy_intrinsic_xyz():
    for e in _angles(2000, 11):
        ours = ingest._euler2rot3d(e)
        theirs = Rotation.from_euler('XYZ', [-e[0], -e[1], e[2]], degrees=True).as_matrix()
        assert np.abs(ours - theirs).max() <= 4 * ULP, (e, ours, theirs)


def test_refshim_euler2mat_matches_scipy_intrinsic_xyz():
    for e in _angles(2000, 12):
        r = np.deg2rad(e)
        ours = refshim._euler2mat(r[0], r[1], r[2], axes='rxyz')
        theirs = Rotation.from_euler('XYZ', r).as_matrix()
        assert np.abs(ours - theirs).max() <= 4 * ULP, (e, ours, theirs)


def test_the_two_restatements_are_the_same_function_bit_for_bit():
    for e in _angles(500, 13):
        a = ingest._euler2rot3d(e)
        r = np.deg2rad(np.array([-e[0], -e[1], e[2]]))
        b = refshim._euler2mat(r[0], r[1], r[2], axes='rxyz')
        assert np.array_equal(a, b)


def test_convention_is_not_any_of_the_lookalikes():
    """the pin would be worthless if the neighbouring conventions also passed"""
    e = np.array([20., -35., 50.])
    ours = ingest._euler2rot3d(e)
    wrong = [Rotation.from_euler('xyz', [-e[0], -e[1], e[2]], degrees=True).as_matrix(),   # extrinsic
             Rotation.from_euler('ZYX', [e[2], -e[1], -e[0]], degrees=True).as_matrix().T,
             Rotation.from_euler('XYZ', [e[0], e[1], e[2]], degrees=True).as_matrix(),     # no sign flip
             Rotation.from_euler('XYZ', [-e[0], -e[1], -e[2]], degrees=True).as_matrix()]
    for w in wrong:
        assert np.abs(ours - w).max() > 1e-3
    # proper rotation, to rounding
    assert abs(np.linalg.det(ours) - 1.) < 1e-14
    assert np.abs(ours @ ours.T - np.eye(3)).max() < 1e-15


@pytest.mark.parametrize('alpha', [10., -45., 90.])
def test_alpha_is_left_handed_about_x(alpha):
    """optical-design convention (euler2opt flips alpha and beta): the local
    +z axis of a surface tilted by +alpha has y component +sin(alpha) in the
    parent frame."""
    r = ingest._euler2rot3d(np.array([alpha, 0., 0.]))
    z = r @ np.array([0., 0., 1.])
    a = math.radians(alpha)
    assert z == pytest.approx([0., math.sin(a), math.cos(a)], abs=1e-15)
