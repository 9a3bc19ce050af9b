"""oracle/operations_oracle.py against vectors produced by the genuine reference."""
import numpy as np
import pytest

from helpers import load
from oracle import operations_oracle as oo


@pytest.fixture(scope="module")
def fx():
    return load("operations.npz")


def test_quadratic_helpers(fx):
    a, b, c = fx["quad_a"], fx["quad_b"], fx["quad_c"]
    with np.errstate(all="ignore"):
        assert np.array_equal(oo.binomial_root(a, b, c), fx["binomial_root"], equal_nan=True)
        assert np.array_equal(oo.smallest_positive_root(a, b, c), fx["smallest_positive_root"], equal_nan=True)
    # test_operations.py:150-164: a = 0 -> the linear root (2); a = b = 0, c < 0 -> (-inf, inf)
    assert np.array_equal(fx["binomial_known"], [[2.0, -np.inf], [2.0, np.inf]])
    assert np.array_equal(oo.binomial_root([0.0], [1.0], [-2.0])[:, 0], (2, 2))
    assert np.array_equal(oo.binomial_root([0.0], [0.0], [-1.0])[:, 0], (-np.inf, np.inf))


def test_dot_products(fx):
    assert np.array_equal(oo.element_wise_dot(fx["dot_m1"], fx["dot_m2"], 0), fx["dot_axis0"])
    assert np.array_equal(oo.element_wise_dot(fx["dot_m1"], fx["dot_m2"], 1), fx["dot_axis1"])
    assert oo.element_wise_dot(fx["dot_m1"][:, 0], fx["dot_m2"][:, 0]) == fx["dot_1d"]


def test_reflect_forms(fx):
    v, n = fx["vectors"], fx["normals"]
    assert np.array_equal(oo.reflect(v, n), fx["reflect_full"])
    assert np.array_equal(oo.reflect(v, n[:, 3]), fx["reflect_one_normal"])
    assert np.array_equal(oo.reflect(v[:, 5], n[:, 5]), fx["reflect_1d"])


def test_refract_forms(fx):
    v, n = fx["vectors"], fx["normals"]
    out, index, unit = oo.refract(v, n, fx["n1"], fx["n2"])
    assert np.array_equal(out, fx["refracted"]) and np.array_equal(index, fx["n_refracted"])
    assert np.array_equal(unit, fx["refract_vectors_after"])
    out, index, _ = oo.refract(v, n, 1.0, 1.5)
    assert np.array_equal(out, fx["refracted_scalar_index"]) and np.array_equal(index, fx["n_refracted_scalar_index"])
    out, index, _ = oo.refract(v, n, fx["n1"], fx["n2"], n_global=1.33)
    assert np.array_equal(out, fx["refracted_world_133"]) and np.array_equal(index, fx["n_refracted_world_133"])
