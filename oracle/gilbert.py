"""ORACLE: ctypes front-end of gilbert_oracle.c (see that file for the reference line map)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgilbert_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.go_xyz2d.restype = ctypes.c_longlong
        L.go_xyz2d.argtypes = [ctypes.c_longlong] * 6
        L.go_mapping.argtypes = [ctypes.c_int] * 3 + [i64p, i64p]
        L.go_sliced_mapping.argtypes = [ctypes.c_int] * 3 + [i64p, i64p]
        L.go_neighbors.argtypes = [ctypes.c_int] * 4 + [i64p, ctypes.POINTER(ctypes.c_uint8)]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def gilbert_xyz2d(x, y, z, w, h, d):
    return int(_lib().go_xyz2d(x, y, z, w, h, d))


def gilbert_mapping(t, h, w):
    """-> (linear_to_hilbert, hilbert_order) int64 arrays (gilbert.py:442-488)."""
    n = t * h * w
    l2h, h2l = np.empty(n, np.int64), np.empty(n, np.int64)
    _lib().go_mapping(t, h, w, _p(l2h, ctypes.c_int64), _p(h2l, ctypes.c_int64))
    return l2h, h2l


def sliced_gilbert_mapping(t, h, w):
    n = t * h * w
    l2h, h2l = np.empty(n, np.int64), np.empty(n, np.int64)
    _lib().go_sliced_mapping(t, h, w, _p(l2h, ctypes.c_int64), _p(h2l, ctypes.c_int64))
    return l2h, h2l


def transpose_gilbert_mapping(dims, order=None):
    """gilbert.py:274-330: the curve of the axis-permuted cuboid.  (t', h', w') = dims[order]; the voxel with coordinates c
    (in the ORIGINAL axis order, linear index = row-major over dims) gets gilbert_xyz2d(c[order[2]], c[order[1]],
    c[order[0]], w', h', t') -- computed point by point with the oracle's own gilbert_xyz2d, as the reference does."""
    if len(dims) != 3:
        raise ValueError("Dimensions must be three-dimensional")
    order = [0, 1, 2] if order is None else list(order)
    if len(order) != 3 or set(order) != {0, 1, 2}:
        raise ValueError("order must be a permutation of 0,1,2")
    t, h, w = (int(dims[o]) for o in order)
    n = int(dims[0]) * int(dims[1]) * int(dims[2])
    l2h, h2l = np.empty(n, np.int64), np.empty(n, np.int64)
    for lin, c in enumerate(np.ndindex(*[int(d) for d in dims])):
        g = gilbert_xyz2d(c[order[2]], c[order[1]], c[order[0]], w, h, t)
        l2h[lin] = g
        h2l[g] = lin
    return l2h, h2l


def block_neighbors(t, h, w, l2h, block_size=128):
    """-> bool [nb, nb] (gilbert.py:597-677; the sliced variant :679-766 differs only in the l2h it colours with)."""
    n = t * h * w
    nb = (n + block_size - 1) // block_size
    out = np.zeros((nb, nb), np.uint8)
    l2h = np.ascontiguousarray(l2h, np.int64)
    _lib().go_neighbors(t, h, w, block_size, _p(l2h, ctypes.c_int64), _p(out, ctypes.c_uint8))
    return out.astype(bool)


def gilbert_block_neighbor_mapping(t, h, w, block_size=128):
    return block_neighbors(t, h, w, gilbert_mapping(t, h, w)[0], block_size)


def sliced_gilbert_block_neighbor_mapping(t, h, w, block_size=128):
    return block_neighbors(t, h, w, sliced_gilbert_mapping(t, h, w)[0], block_size)
