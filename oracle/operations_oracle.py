"""CPU oracle for tinygfx/g3d/operations.py as callable functions (rows a5 / a12 of SURVEY.md
section 8a): smallest_positive_root :4-25, binomial_root :28-63, element_wise_dot :66-83,
reflect :86-107, refract :110-162, in every argument form upstream accepts.

TEST INFRASTRUCTURE ONLY (same rule as prt_oracle).  PINNED by tests/golden/operations.npz, which
the genuine reference produced (tests/golden/generate_golden.py operations)."""
import numpy as np

from . import prt_oracle as po

INF = np.inf


def binomial_root(a, b, c):
    return po.quadratic_pair(np.asarray(a, float), np.asarray(b, float), np.asarray(c, float))


def smallest_positive_root(a, b, c):
    """The smaller non-negative root, +inf when there is none; a == 0 is only protected against
    the division (upstream leaves filtering such entries to the caller)."""
    a, b, c = (np.asarray(v, float) for v in (a, b, c))
    disc = b ** 2 - 4 * a * c
    s = np.sqrt(np.maximum(0, disc))
    roots = np.vstack((-b + s, -b - s)) / (2 * a + po.near_zero(a))
    pick = np.where(roots[1] >= 0, np.minimum(roots[0], roots[1]), roots[0])
    return np.where((disc >= 0) & (pick >= 0), pick, INF)


def element_wise_dot(m1, m2, axis=0):
    m1, m2 = np.asarray(m1, float), np.asarray(m2, float)
    if m1.ndim == 1:
        return m1.dot(m2)
    return np.einsum("ij,ij->j" if axis == 0 else "ij,ij->i", m1, m2)


def reflect(vectors, normals):
    vectors, normals = np.ascontiguousarray(vectors, float), np.ascontiguousarray(normals, float)
    if vectors.ndim == 1 and normals.ndim == 1:
        return vectors - normals * 2 * vectors.dot(normals)
    if normals.ndim == 1:
        along = np.einsum("ij,i->j", vectors, normals)
        return vectors - 2 * np.tile(normals, (vectors.shape[1], 1)).T * along
    return po.reflect(vectors, normals)


def refract(vectors, normals, n1, n2, n_global=1):
    """Returns (refracted, n_refracted, vectors normalised) -- upstream normalises its first
    argument in place, which callers can observe."""
    vectors = np.asarray(vectors, float)
    unit = vectors / np.linalg.norm(vectors, axis=0)
    out, index = po.refract(vectors, np.asarray(normals, float), n1, n2, n_global)
    return out, index, unit
