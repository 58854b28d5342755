"""Shared test helpers (conversion between torch low-precision tensors and the oracle's fp32 arrays)."""
import threading

import numpy as np
import torch


class SimWorld:
    """Shared state of N simulated ranks (threads of one process, one GPU)."""

    def __init__(self, N):
        self.N = N
        self.barrier = threading.Barrier(N)
        self.slots = {}
        self.lock = threading.Lock()


class _SimParty:
    def __init__(self, world, rank, tag):
        self.w, self.r, self.calls, self.tag = world, rank, 0, tag

    def size(self):
        return self.w.N

    def rank(self):
        return self.r

    def _swap(self, payload):
        key = (self.tag, self.calls)
        self.calls += 1
        with self.w.lock:
            self.w.slots[(key, self.r)] = payload
        self.w.barrier.wait()
        peers = [self.w.slots[(key, p)] for p in range(self.w.N)]
        return key, peers

    def _done(self, key):
        self.w.barrier.wait()            # everybody has copied: the send buffers may go
        with self.w.lock:
            self.w.slots.pop((key, self.r), None)


class _Waited:
    def wait(self):
        return True


class SimExchange(_SimParty):
    """In-process stand-in for jenga_amd.modules.ulysses.DistExchange: same methods, the chunks REALLY change ranks
    (recv[r][p] = send[p][r], what all_to_all_single / the grouped send-recv do)."""

    def __init__(self, world, rank):
        super().__init__(world, rank, "x")

    def all_to_all(self, recvs, sends):
        key, peers = self._swap(sends)
        for i, rc in enumerate(recvs):
            for p in range(self.w.N):
                rc[p].copy_(peers[p][i][self.r])       # chunk r of rank p's send buffer -> chunk p of my receive buffer
        self._done(key)
        return [_Waited()]

    def all_gather(self, out, x):
        key, peers = self._swap(x)
        for p in range(self.w.N):
            out[p].copy_(peers[p])
        self._done(key)
        return _Waited()


class SimGroup(_SimParty):
    """In-process stand-in for the sequence-parallel group object (ulysses._SPGroup): .all_gather(x, dim) concatenates
    the ranks' tensors in rank order along `dim` (jenga_hyvideo_multigpu.py:193)."""

    def __init__(self, world, rank):
        super().__init__(world, rank, "g")

    def all_gather(self, x, dim=0):
        key, peers = self._swap(x)
        out = torch.cat(list(peers), dim=dim)   # (all threads share the device's default stream: ordered)
        self._done(key)
        return out


def to_np(t):
    """torch tensor (bf16/fp16/fp32) -> float32 numpy array with the same values."""
    return t.detach().float().cpu().numpy()


def from_bits(a, dtype):
    """golden arrays: bf16 stored as uint16 bit patterns, fp16 stored natively."""
    if dtype in ("bfloat16", "bf16"):
        return torch.from_numpy(a.astype(np.uint16)).view(torch.bfloat16)
    return torch.from_numpy(a)


def tie_tolerant_mask_equal(mask, ref_mask, probs, n, nb_img, forced):
    """The reference's torch.sort is unstable, so inside a group of equal probabilities ANY members may be the
    ones selected.  Accept `mask` iff, per row: columns forced by neighbours/text/first-frame agree, the count of
    sort-selected-or-forced columns is consistent, and no unselected image column has a strictly larger
    probability than a selected non-forced one.  Returns (ok, message)."""
    B, H, nq, _ = mask.shape
    if not np.array_equal(mask[..., nb_img:], ref_mask[..., nb_img:]):
        return False, "text columns differ"
    for b in range(B):
        for h in range(H):
            for r in range(nq):
                m, rm = mask[b, h, r, :nb_img], ref_mask[b, h, r, :nb_img]
                f = forced[b, h, r, :nb_img]
                if not np.all(m[f]) or not np.all(rm[f]):
                    return False, f"forced column missing at {(b, h, r)}"
                pr = probs[b, h, r]
                k = int(min(n[b, h, r], nb_img))
                cutoff = np.sort(pr)[::-1][k - 1]
                # everything strictly above the cutoff must be in, everything strictly below must be out unless forced
                if not np.all(m[pr > cutoff]):
                    return False, f"row {(b, h, r)}: a block above the cutoff is missing"
                if np.any(m[(pr < cutoff) & ~f]):
                    return False, f"row {(b, h, r)}: a block below the cutoff was selected"
                # number taken from the tie group must match what n requires
                need = k - int((pr > cutoff).sum())
                tie = (pr == cutoff)
                got_min = int((m & tie & ~f).sum())
                if int((m & tie).sum()) < min(need, int(tie.sum())) or got_min > need:
                    return False, f"row {(b, h, r)}: wrong number of tie-group members"
    return True, ""


def assert_ulp_close(a, b, dtype, max_frac=1e-3, max_ulps=1, rowwise=False):
    """a, b float32 arrays of `dtype`-representable values: equal except on <= max_frac of the elements, where they
    may differ by <= max_ulps units in the last place.  rowwise=True measures the ulp at the largest magnitude of the
    last axis (RoPE mixes the two members of a pair, so an input flip shows up scaled by the larger partner)."""
    mant = 7 if dtype in ("bfloat16", "bf16") else 10
    diff = a != b
    frac = diff.mean()
    assert frac <= max_frac, f"{frac:.2e} of elements differ"
    if diff.any():
        mag = np.maximum(np.abs(a), np.abs(b))
        if rowwise:
            mag = np.broadcast_to(mag.max(axis=-1, keepdims=True), a.shape)
        ulp = np.exp2(np.floor(np.log2(np.maximum(mag[diff], 1e-30))) - mant)
        worst = (np.abs(a[diff] - b[diff]) / ulp).max()
        assert worst <= max_ulps * 1.001, f"difference of {worst:.2f} ulp (allowed {max_ulps})"


def hip_pooled(q, k, text_blocks, dev, bf16=False):
    """The block means the AttenCarve op selects from, computed by the HIP pooling kernel exactly as the op does it (q / k
    [B,S,H,128] torch tensors, zero-padded to whole 128-token blocks like the padding flavours do): -> (qp [B,H,nimg,128],
    kp [B,H,nb,128]) as fp32 numpy arrays for oracle.attention.block_sparse_attention(pooled=...).  With them the oracle
    selects from the SAME pooled values as the kernel under test, so the whole-op comparisons can demand every row."""
    from jenga_amd import _capi
    B, S, H, D = q.shape
    pad = (128 - S % 128) % 128
    qd, kd = q.to(dev), k.to(dev)
    if bf16:
        qd, kd = qd.to(torch.bfloat16), kd.to(torch.bfloat16)
    if pad:
        qd = torch.nn.functional.pad(qd, [0, 0, 0, 0, 0, pad])
        kd = torch.nn.functional.pad(kd, [0, 0, 0, 0, 0, pad])
    nb = (S + pad) // 128
    nimg = nb - text_blocks
    if nimg <= 0:
        return None
    qp = _capi.block_pool(qd.contiguous(), nimg)
    kp = _capi.block_pool(kd.contiguous(), nb)
    torch.cuda.synchronize()
    return qp.float().cpu().numpy(), kp.float().cpu().numpy()


# ---- multi-process runs on ONE GPU (tests/test_gpu_rccl.py): the exchange goes through the host on a gloo group -------------
class _HostDone:
    def wait(self):
        return True


class HostStagedExchange:
    """Test infrastructure: the interface of jenga_amd.modules.ulysses.DistExchange (all_to_all / all_gather on [N, ...]
    buffers) on a gloo process group, every buffer staged through host memory -- so that N real processes that SHARE one GPU
    (RCCL refuses two ranks on one device) can run the product's sequence-parallel path with its HIP local steps."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.bytes_out = 0
        self.calls = 0

    def size(self):
        return self.dist.get_world_size(self.group)

    def rank(self):
        return self.dist.get_rank(self.group)

    def all_to_all(self, recvs, sends):
        self.calls += 1
        for rc, sd in zip(recvs, sends):
            torch.cuda.synchronize(sd.device)
            h_in = sd.contiguous().cpu()
            h_out = torch.empty_like(h_in)
            self.dist.all_to_all_single(h_out, h_in, group=self.group)
            rc.copy_(h_out.to(rc.device))
        return [_HostDone()]

    def all_gather(self, out, x):
        self.calls += 1
        torch.cuda.synchronize(x.device)
        h = x.contiguous().cpu()
        parts = [torch.empty_like(h) for _ in range(self.size())]
        self.dist.all_gather(parts, h, group=self.group)
        out.copy_(torch.stack(parts, 0).to(out.device).view_as(out))
        return _HostDone()


class HostStagedGroup:
    """The group object of jenga_amd.modules.ulysses.set_thread_sp_group (.size, .rank, .all_gather(x, dim), .group) on gloo."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group if group is not None else dist.group.WORLD

    def size(self):
        return self.dist.get_world_size(self.group)

    def rank(self):
        return self.dist.get_rank(self.group)

    def all_gather(self, x, dim=0):
        torch.cuda.synchronize(x.device)
        h = x.contiguous().cpu()
        parts = [torch.empty_like(h) for _ in range(self.size())]
        self.dist.all_gather(parts, h, group=self.group)
        return torch.cat(parts, dim=dim).to(x.device)
