"""Vector and polynomial helpers of the reference's ``tinygfx/g3d/operations.py``, as functions.

Same names, argument forms and return shapes as upstream -- ``smallest_positive_root`` (``:4-25``),
``binomial_root`` (``:28-63``), ``element_wise_dot`` (``:66-83``), ``reflect`` (``:86-107``),
``refract`` (``:110-162``) -- evaluated by the device functions the trace itself is built from
(``binomial_root``, ``reflect4``, ``refract4`` in ``csrc/prt_device.hpp``) through the ``prt_*``
entry points of ``include/prt.h``.  Host arrays go in and come out, like every object-level call
of this package; there is no host implementation behind them.
"""
import numpy as np

__all__ = ["smallest_positive_root", "binomial_root", "element_wise_dot", "reflect", "refract"]


def _engine():
    from .. import engine

    return engine


def _columns(array):
    """(m,) or (m, n) -> contiguous float64 (m, n), and whether it was 1-D."""
    a = np.asarray(array, dtype=float)
    single = a.ndim == 1
    return np.ascontiguousarray(a.reshape(a.shape[0], -1)), single


def _coefficients(*values):
    arrays = np.broadcast_arrays(*(np.atleast_1d(np.asarray(v, dtype=float)) for v in values))
    return [np.array(a) for a in arrays]


def smallest_positive_root(a, b, c):
    """Smallest root >= 0 of a x^2 + b x + c, +inf where there is none."""
    return _engine().ops_polynomial("prt_smallest_positive_root", *_coefficients(a, b, c))


def binomial_root(a, b, c, disc=None):
    """(2, n) roots of a x^2 + b x + c with upstream's handling of the linear (|a| <= 1e-8) and
    constant (|a|, |b| <= 1e-8 -> -inf / +inf by the sign of c) cases; no real root -> +inf."""
    if disc is not None:
        raise NotImplementedError("a precomputed discriminant is not supported; it is recomputed on the device")
    return _engine().ops_polynomial("prt_binomial_root", *_coefficients(a, b, c))


def element_wise_dot(mat_1, mat_2, axis=0):
    """Column-wise (axis 0) or row-wise (axis 1) dot products of two (m, n) blocks; a plain dot
    product for 1-D arguments."""
    m1, single = _columns(mat_1)
    m2, _ = _columns(mat_2)
    if m1.shape != m2.shape:
        raise ValueError("operands must have the same shape")
    out = _engine().ops_dot(m1, m2, 0 if single else axis)
    return out[0] if single else out


def reflect(vectors, normals):
    """v - 2 n (v.n) for (m, n) column vectors; ``normals`` may be a single (m,) normal."""
    v, single = _columns(vectors)
    nrm, one_normal = _columns(normals)
    if one_normal and not single:
        nrm = np.array(np.broadcast_to(nrm, v.shape))
    if nrm.shape != v.shape:
        raise ValueError("vectors and normals must have the same shape")
    out = _engine().ops_reflect(v, nrm)
    return out[:, 0] if single else out


def refract(vectors, normals, n1, n2, n_global=1):
    """Vector form of Snell's law with exit detection and total internal reflection: returns
    (new unit directions, refractive index the ray now travels in).  ``vectors`` is normalised in
    place, as upstream does."""
    v, single = _columns(vectors)
    nrm, _ = _columns(normals)
    if nrm.shape != v.shape:
        raise ValueError("vectors and normals must have the same shape")
    count = v.shape[1]
    index_in = np.array(np.broadcast_to(np.asarray(n1, dtype=float), (count,)))
    index_behind = np.array(np.broadcast_to(np.asarray(n2, dtype=float), (count,)))
    out, index, unit = _engine().ops_refract(v, nrm, index_in, index_behind, float(n_global))
    if isinstance(vectors, np.ndarray) and vectors.dtype == np.float64:
        vectors[...] = unit[:, 0] if single else unit
    return (out[:, 0], index[0]) if single else (out, index)
