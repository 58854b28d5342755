"""Generalized-Hilbert ("Gilbert") token reorder, drop-in for the reference's gilbert.py.

Same function names and return conventions as the reference (gilbert_mapping :442-488, sliced_gilbert_mapping
:332-440, gilbert_block_neighbor_mapping :597-677, sliced_gilbert_block_neighbor_mapping :679-766): the mapping
functions return two Python lists (linear_to_hilbert, hilbert_order), the neighbour functions a CPU torch.bool
[nb, nb] tensor -- but the work is done by one HIP kernel launch (one thread per voxel) instead of ~10 s of Python
recursion.  Pass `as_tensor=True` to keep the results on the GPU (what jenga_amd's own driver does).
"""
import torch

from . import _capi


def _dev(device):
    if device is None:
        if not torch.cuda.is_available():
            raise _capi.JengaError("jenga_amd.gilbert needs a GPU (no CPU fallback)")
        device = torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def _mapping(t, h, w, sliced, transpose_order, as_tensor, device):
    if transpose_order is not None:
        raise NotImplementedError("transpose_order is unused by every Jenga entry script and is not supported")
    l2h, h2l = _capi.gilbert_map(int(t), int(h), int(w), sliced, _dev(device))
    if as_tensor:
        return l2h, h2l
    return l2h.cpu().tolist(), h2l.cpu().tolist()


def gilbert_mapping(t, h, w, transpose_order=None, as_tensor=False, device=None):
    return _mapping(t, h, w, False, transpose_order, as_tensor, device)


def sliced_gilbert_mapping(t, h, w, transpose_order=None, as_tensor=False, device=None):
    return _mapping(t, h, w, True, transpose_order, as_tensor, device)


def _neighbors(t, h, w, block_size, sliced, transpose_order, as_tensor, device):
    if transpose_order is not None:
        raise NotImplementedError("transpose_order is not supported")
    l2h, _ = _capi.gilbert_map(int(t), int(h), int(w), sliced, _dev(device))
    nb = _capi.gilbert_neighbors(int(t), int(h), int(w), int(block_size), l2h)
    return nb if as_tensor else nb.cpu()


def gilbert_block_neighbor_mapping(t, h, w, block_size=128, transpose_order=None, as_tensor=False, device=None):
    return _neighbors(t, h, w, block_size, False, transpose_order, as_tensor, device)


def sliced_gilbert_block_neighbor_mapping(t, h, w, block_size=128, transpose_order=None, as_tensor=False,
                                          device=None):
    return _neighbors(t, h, w, block_size, True, transpose_order, as_tensor, device)


def transpose_gilbert_mapping(dims, order=None, as_tensor=False, device=None):
    """gilbert.py:274-330 (imported by jenga_hyvideo.py:24 / jenga_hyi2v.py:26, called by no entry script): with the
    default axis order it is gilbert_mapping(*dims); other orders are not supported."""
    if len(dims) != 3:
        raise ValueError("Dimensions must be three-dimensional")
    if order is not None and list(order) != [0, 1, 2]:
        if len(order) != 3 or set(order) != {0, 1, 2}:
            raise ValueError("order must be a permutation of 0,1,2")
        raise NotImplementedError("only the default axis order [0, 1, 2] is supported (no Jenga entry script passes another)")
    return _mapping(dims[0], dims[1], dims[2], False, None, as_tensor, device)
