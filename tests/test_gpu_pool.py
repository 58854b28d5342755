"""GPU (-m gpu): the 128-token block means the selection starts from, DIRECTLY against the oracle's pooling formula at the
HunyuanVideo 720p shape (900 image + 2 text blocks, 24 heads).

Reference: `q.reshape(B, H, nb, 128, D).mean(dim=-2)` on a 16-bit tensor (attention_block_triton_diffres.py:216-217) = the fp32
mean rounded ONCE to the dtype; oracle/attention.py::pooled_scores restates it as rnd(mean_fp32(x)).  Until round 5 the pooled
values were only checked through the masks they lead to (small cases) and as "fused == separate" (a self-comparison); every
whole-op test hands the oracle the HIP pooled means.  This file closes that: `jenga_block_pool` and the pooled output of
`jenga_qk_norm_rope_pool` against rnd(fp32 mean) computed by numpy from the same rows, in ulps.  The only freedom is the fp32
summation order of 128 addends, i.e. a result on a rounding boundary may land on the neighbouring 16-bit value: <= 1 ulp on
<= 0.2 % of the values, everything else bit-equal (recorded in gpurun_out/parity_records/pool_full_size.json)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import assert_ulp_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NIMG, NTXT, H = 900, 2, 24


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _oracle_pool(x_bshd, n_blocks, dtype_name):
    """rnd(mean over 128-token blocks in fp32) of x [1,S,H,128] (a torch tensor on the device) -> fp32 numpy [1,H,n,128];
    head by head on the host (numpy pairwise summation, the oracle's own formula)."""
    from oracle.rounding import rounder
    rnd = rounder(dtype_name)
    out = np.empty((1, x_bshd.shape[2], n_blocks, 128), np.float32)
    for h in range(x_bshd.shape[2]):
        xh = x_bshd[0, : n_blocks * 128, h].float().cpu().numpy()
        out[0, h] = rnd(xh.reshape(n_blocks, 128, 128).mean(axis=1, dtype=np.float32))
    return out


def _record(tag, got, ref):
    diff = got != ref
    rec = {tag: dict(values=int(ref.size), differing=int(diff.sum()), frac=float(diff.mean()))}
    try:
        d = os.path.join(ROOT, "gpurun_out", "parity_records")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "pool_full_size.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data.update(rec)
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    print("PARITY-RECORD pool_full_size.json", json.dumps(rec))


@pytest.mark.parametrize("dt,name", [(torch.bfloat16, "bfloat16"), (torch.float16, "float16")])
def test_block_pool_vs_oracle_at_720p(dev, dt, name):
    from jenga_amd import _capi
    g = torch.Generator(device=dev).manual_seed(31)
    nb = NIMG + NTXT
    # block-dependent offsets: the means are not all ~0 (rounding boundaries at many exponents)
    x = torch.randn(1, nb * 128, H, 128, generator=g, device=dev)
    x += torch.randn(1, nb, 1, H, 128, generator=g, device=dev).expand(1, nb, 128, H, 128).reshape(1, nb * 128, H, 128) * 2.0
    x = x.to(dt)
    for n in (NIMG, nb):           # Q pools the image blocks, K all blocks
        got = _capi.block_pool(x, n).float().cpu().numpy()
        ref = _oracle_pool(x, n, name)
        _record(f"block_pool {name} n={n}", got, ref)
        assert_ulp_close(got, ref, name, max_frac=2e-3, max_ulps=1)


def test_fused_norm_rope_pool_pooled_output_vs_oracle_at_720p(dev):
    """The pooled half of jenga_qk_norm_rope_pool: means of the kernel's OWN normed + rotated 16-bit rows (those rows are pinned
    to the reference goldens by test_gpu_parity.py::test_rmsnorm_rope_vs_reference_golden and test_gpu_fused.py)."""
    from jenga_amd import _capi
    g = torch.Generator(device=dev).manual_seed(32)
    nb = NIMG + NTXT
    S, S_img = nb * 128, NIMG * 128
    lin = (torch.randn(1, S, 3 * H * 128, generator=g, device=dev) * 1.3).to(torch.bfloat16)
    qkv = lin.unflatten(-1, (3, H, 128))
    xq, xk = qkv[:, :, 0], qkv[:, :, 1]
    wq = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(torch.bfloat16)
    wk = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(torch.bfloat16)
    ang = torch.rand(S_img, 64, generator=g, device=dev) * 6.28
    cos = torch.cos(ang).repeat_interleave(2, dim=1).contiguous()
    sin = torch.sin(ang).repeat_interleave(2, dim=1).contiguous()
    oq = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=dev)
    ok = torch.empty_like(oq)
    qp = torch.zeros(1, H, NIMG, 128, dtype=torch.bfloat16, device=dev)
    kp = torch.zeros(1, H, nb, 128, dtype=torch.bfloat16, device=dev)
    _capi.qk_norm_rope_pool(xq, xk, wq, wk, cos, sin, oq, ok, s_rope=S_img, qpool=qp, kpool=kp)
    torch.cuda.synchronize()
    for tag, rows, pooled, n in (("q", oq, qp, NIMG), ("k", ok, kp, nb)):
        got = pooled.float().cpu().numpy()
        ref = _oracle_pool(rows, n, "bfloat16")
        _record(f"qk_norm_rope_pool pooled {tag}", got, ref)
        assert_ulp_close(got, ref, "bfloat16", max_frac=2e-3, max_ulps=1)
