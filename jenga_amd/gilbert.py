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
        # gilbert.py:436-438, 484-486: with an axis order BOTH mappings are the transposed 3-D curve (the sliced variant too)
        return transpose_gilbert_mapping([t, h, w], transpose_order, as_tensor=as_tensor, device=device)
    l2h, h2l = _capi.gilbert_map(int(t), int(h), int(w), sliced, _dev(device))
    if as_tensor:
        return l2h, h2l
    return l2h.cpu().tolist(), h2l.cpu().tolist()


def gilbert_mapping(t, h, w, transpose_order=None, as_tensor=False, device=None):
    return _mapping(t, h, w, False, transpose_order, as_tensor, device)


def sliced_gilbert_mapping(t, h, w, transpose_order=None, as_tensor=False, device=None):
    return _mapping(t, h, w, True, transpose_order, as_tensor, device)


def _neighbors(t, h, w, block_size, sliced, transpose_order, as_tensor, device):
    # (transpose_order is accepted and ignored, as in the reference: gilbert.py:597-677 / 679-766 never read it)
    l2h, _ = _capi.gilbert_map(int(t), int(h), int(w), sliced, _dev(device))
    nb = _capi.gilbert_neighbors(int(t), int(h), int(w), int(block_size), l2h)
    return nb if as_tensor else nb.cpu()


def gilbert_block_neighbor_mapping(t, h, w, block_size=128, transpose_order=None, as_tensor=False, device=None):
    return _neighbors(t, h, w, block_size, False, transpose_order, as_tensor, device)


def sliced_gilbert_block_neighbor_mapping(t, h, w, block_size=128, transpose_order=None, as_tensor=False,
                                          device=None):
    return _neighbors(t, h, w, block_size, True, transpose_order, as_tensor, device)


def transpose_gilbert_mapping(dims, order=None, as_tensor=False, device=None):
    """gilbert.py:274-330: the curve of the axis-permuted cuboid.  (t', h', w') = dims[order]; the voxel with coordinates c
    in the original axis order gets gilbert_xyz2d(c[order[2]], c[order[1]], c[order[0]], w', h', t').  The HIP kernel
    computes the curve of the (t', h', w') cuboid (one thread per voxel); the axis permutation of the result is an index
    view (arange.view(t', h', w').permute(...)) -- integer plumbing, no arithmetic."""
    if len(dims) != 3:
        raise ValueError("Dimensions must be three-dimensional")
    order = [0, 1, 2] if order is None else [int(o) for o in order]
    if len(order) != 3 or set(order) != {0, 1, 2}:
        raise ValueError("order must be a permutation of 0,1,2")
    if order == [0, 1, 2]:
        return _mapping(dims[0], dims[1], dims[2], False, None, as_tensor, device)
    dev = _dev(device)
    tp, hp, wp = (int(dims[o]) for o in order)
    l2h_t, _ = _capi.gilbert_map(tp, hp, wp, False, dev)          # [z' h' w' + y' w' + x'] -> gilbert(x', y', z'; w', h', t')
    n = tp * hp * wp
    inv = [order.index(k) for k in range(3)]                       # original axis k is axis inv[k] of the permuted cuboid
    src = torch.arange(n, device=dev).view(tp, hp, wp).permute(*inv).reshape(-1)
    l2h = l2h_t[src].contiguous()
    h2l = torch.empty_like(l2h)
    h2l[l2h] = torch.arange(n, device=dev, dtype=l2h.dtype)
    if as_tensor:
        return l2h, h2l
    return l2h.cpu().tolist(), h2l.cpu().tolist()
