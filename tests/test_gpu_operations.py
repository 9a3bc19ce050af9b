"""pyrayt_amd.g3d.operations (tinygfx/g3d/operations.py as functions, on the device) against
vectors produced by the genuine reference (tests/golden/operations.npz)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ATOL = helpers.ATOL


@pytest.fixture(scope="module")
def fx():
    return helpers.load("operations.npz")


@pytest.fixture(scope="module")
def cg():
    import pyrayt_amd.g3d as g3d

    return g3d


def same(got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    finite = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), finite) and np.array_equal(got[~finite], want[~finite], equal_nan=True)
    assert np.allclose(got[finite], want[finite], rtol=0, atol=ATOL)
    return float(np.abs(got[finite] - want[finite]).max()) if finite.any() else 0.0


def test_quadratic_helpers(fx, cg):
    a, b, c = fx["quad_a"], fx["quad_b"], fx["quad_c"]
    assert same(cg.binomial_root(a, b, c), fx["binomial_root"]) == 0.0
    assert same(cg.smallest_positive_root(a, b, c), fx["smallest_positive_root"]) == 0.0
    # test_operations.py:150-164
    assert np.array_equal(cg.binomial_root(np.array([0.0]), np.array([1.0]), np.array([-2.0]))[:, 0], (2, 2))
    assert np.array_equal(cg.binomial_root(0.0, 0.0, -1.0)[:, 0], (-np.inf, np.inf))
    with pytest.raises(NotImplementedError):
        cg.binomial_root(a, b, c, disc=b * b)


def test_dot_products(fx, cg):
    assert same(cg.element_wise_dot(fx["dot_m1"], fx["dot_m2"], axis=0), fx["dot_axis0"]) == 0.0
    same(cg.element_wise_dot(fx["dot_m1"], fx["dot_m2"], axis=1), fx["dot_axis1"])  # numpy sums rows pairwise
    assert abs(cg.element_wise_dot(fx["dot_m1"][:, 0], fx["dot_m2"][:, 0]) - fx["dot_1d"]) < 1e-12


def test_reflect_forms(fx, cg):
    v, n = fx["vectors"], fx["normals"]
    assert same(cg.reflect(v, n), fx["reflect_full"]) == 0.0
    same(cg.reflect(v, n[:, 3]), fx["reflect_one_normal"])
    one = cg.reflect(v[:, 5], n[:, 5])
    assert one.shape == (4,)
    same(one, fx["reflect_1d"])


def test_refract_forms(fx, cg):
    v, n = fx["vectors"].copy(), fx["normals"]
    out, index = cg.refract(v, n, fx["n1"], fx["n2"])
    assert same(out, fx["refracted"]) == 0.0 and np.array_equal(index, fx["n_refracted"])
    assert np.array_equal(v, fx["refract_vectors_after"])          # normalised in place, like upstream
    out, index = cg.refract(fx["vectors"].copy(), n, 1.0, 1.5)
    assert same(out, fx["refracted_scalar_index"]) == 0.0 and np.array_equal(index, fx["n_refracted_scalar_index"])
    out, index = cg.refract(fx["vectors"].copy(), n, fx["n1"], fx["n2"], n_global=1.33)
    assert same(out, fx["refracted_world_133"]) == 0.0 and np.array_equal(index, fx["n_refracted_world_133"])
    # the analytic cases of test_operations.py:222-285: 45 degrees into n = 1.5, and TIR leaving it
    ray = np.array((0.0, 1.0, -1.0, 0.0))
    normal = np.array((0.0, 0.0, 1.0, 0.0))
    out, index = cg.refract(ray, normal, 1.0, 1.5)
    assert index == 1.5 and np.isclose(np.arctan(abs(out[1] / out[2])), np.arcsin(np.sin(np.pi / 4) / 1.5))
    out, index = cg.refract(np.array((0.0, 1.0, 1.0, 0.0)), normal, 1.5, 1.5)
    assert index == 1.5 and np.allclose(out, np.array((0, 1, -1, 0)) / np.sqrt(2))


def test_shape_errors(cg):
    with pytest.raises(ValueError):
        cg.reflect(np.zeros((4, 3)), np.zeros((4, 2)))
    with pytest.raises(ValueError):
        cg.element_wise_dot(np.zeros((4, 3)), np.zeros((3, 3)))
    with pytest.raises(ValueError):
        cg.reflect(np.zeros((5, 3)), np.zeros((5, 3)))


def test_array_csg_matches_reference():
    """csg.array_csg on the device: the reference's own known answers (test_csg.py:212-231), random
    lists of every size pairing the fixtures hold, 1-D and unsorted forms."""
    from oracle import prt_oracle as orc
    from pyrayt_amd.g3d import csg

    gold = helpers.load("csg.npz")
    a1, a2 = gold["array_csg__a1"], gold["array_csg__a2"]
    for op in csg.Operation:
        got = csg.array_csg(a1, a2, op)
        assert got.shape == gold[f"array_csg__{op.name}"].shape
        assert np.array_equal(got, gold[f"array_csg__{op.name}"])
    sizes = sorted({k.split("__")[0] for k in gold if k.startswith("array_csg_rand_")})
    assert len(sizes) >= 5
    for key in sizes:
        left, right = gold[key + "__left"], gold[key + "__right"]
        for op in csg.Operation:
            assert np.array_equal(csg.array_csg(left, right, op), gold[f"{key}__{op.name}"]), (key, op)
            values, _ = orc.merge_lists(left, right, op.value)
            assert np.array_equal(csg.array_csg(left, right, op, sort_output=False), values), (key, op)
    with pytest.raises(ValueError):
        csg.array_csg(a1, a2, 7)
    with pytest.raises(ValueError):
        csg.array_csg(np.zeros((3, 4)), np.zeros((2, 4)), csg.Operation.UNION)   # odd list length
