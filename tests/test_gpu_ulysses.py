"""GPU (-m gpu): the Ulysses path with N > 1 on ONE GPU.

(i)  the HIP head pack / unpack kernels (jenga_ulysses_pack_heads / _unpack_heads) for N in {2, 4, 8}, strided inputs,
     bit-exact against the oracle's permutes;
(ii) the whole UlyssesAttenCarve.forward for N in {2, 8}: N simulated ranks run the shipped forward -- HIP pack,
     selection, attention, unpack -- in N threads of one process, joined by an in-process exchange that REALLY permutes
     the chunks between the ranks (recv[r][p] = send[p][r], what all_to_all_single / the grouped send-recv do; those
     collectives themselves are exercised over gloo in tests/test_ulysses_gloo.py).  Every rank's result must equal,
     bit for bit, the single-rank HIP op over all heads at top_k = N * int(...) (heads are independent), and agree
     with the oracle's N-rank simulation within the attention tolerance."""
import threading

import numpy as np
import pytest
import torch

import inputs
from helpers import SimExchange, SimWorld, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("N", [2, 4, 8])
def test_pack_unpack_heads_vs_oracle(dev, N):
    from jenga_amd import _capi
    from oracle import ulysses as ou
    gen = torch.Generator().manual_seed(N)
    B, S_loc, H, D = 1, 144, 8, 128
    # strided input: the q slice of a fused QKV GEMM output [B,S,3,H,D]
    qkv = torch.randn(B, S_loc, 3, H, D, generator=gen).to(torch.bfloat16)
    for which in range(3):
        x = qkv[:, :, which]
        got = _capi.ulysses_pack_heads(qkv.to(dev)[:, :, which], N)
        want = ou.pack_heads(x, N)
        assert got.shape == (N, B, S_loc, H // N, D) and torch.equal(got.cpu(), want)
        # unpack into a strided destination (the image rows of a [B, S_loc + S_txt, H, D] result)
        res = torch.zeros(B, S_loc + 128, H, D, dtype=torch.bfloat16, device=dev)
        _capi.ulysses_unpack_heads(got, N, out=res[:, :S_loc])
        assert torch.equal(res[:, :S_loc].cpu(), x) and not res[:, S_loc:].any()
    # fp16 and a second shape
    y = torch.randn(1, 40, 24, 128, generator=gen).to(torch.float16)
    if 24 % N == 0:
        assert torch.equal(_capi.ulysses_pack_heads(y.to(dev), N).cpu(), ou.pack_heads(y, N))


@pytest.mark.parametrize("N,pipeline", [(2, False), (8, False), (2, True)])
def test_simulated_ranks_forward_equals_single_rank_op(dev, N, pipeline):
    # pipeline=True: what JENGA_ULYSSES_PIPELINE=1 makes of every UlyssesAttenCarve -- the reference-signature call has no
    # head-group pipeline and must run as one group instead of raising
    from jenga_amd.modules import ulysses
    from jenga_amd.modules.attention import my_parallel_attention
    from jenga_amd.modules.attention_block_sparse import block_sparse_attention
    from oracle import gilbert as og
    from oracle import ulysses as ou
    gen = torch.Generator().manual_seed(321 + N)
    H, nimg, tb = 8, 9, 2                     # 1152 image tokens: S_loc = 576 / 144 (144 is NOT a multiple of 128)
    q, k = inputs.peaky_qk(gen, 1, H, nimg + tb, nimg + tb, 128, 0.8)
    q = q.transpose(1, 2).to(torch.bfloat16).contiguous()      # [1,S,H,D]
    k = k.transpose(1, 2).to(torch.bfloat16).contiguous()
    v = torch.randn(1, (nimg + tb) * 128, H, 128, generator=gen).to(torch.bfloat16)
    nbm = og.gilbert_block_neighbor_mapping(3, 12, 32, 128)    # 1152 voxels -> 9 blocks
    assert nbm.shape == (nimg, nimg)
    S_img, S_txt = nimg * 128, tb * 128
    S_loc = S_img // N
    n_valid, amp, p_rate = 70, 0.25, 0.3
    top_k = N * int((1 - 0.5) * (S_loc // 128))                # models_mul...:249-251 on the LOCAL block count
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    nb_dev = torch.from_numpy(nbm).to(dev)

    world = SimWorld(N)
    results, errors = [None] * N, []

    def run(rank):
        try:
            torch.cuda.set_device(dev)
            sl = slice(rank * S_loc, (rank + 1) * S_loc)
            loc = lambda t: torch.cat([t[:, sl], t[:, S_img:]], dim=1)
            cu = torch.tensor([0, S_loc + n_valid, S_loc + S_txt], dtype=torch.int32, device=dev)
            sp = ulysses.UlyssesAttenCarve(exchange=SimExchange(world, rank), pipeline=pipeline)
            out = my_parallel_attention(sp, loc(qd), loc(kd), loc(vd), img_q_len=S_loc, img_kv_len=S_loc,
                                        cu_seqlens_q=cu, cu_seqlens_kv=cu, top_k=top_k, text_amp=amp,
                                        block_neighbor_list=nb_dev, p_remain_rates=p_rate)
            results[rank] = out.reshape(1, S_loc + S_txt, H, 128)
        except Exception as e:                                 # noqa: BLE001 - surfaced below
            errors.append((rank, repr(e)))
            world.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    torch.cuda.synchronize()

    # (a) the single-rank HIP op over all heads with the same top_k: bit for bit
    cu1 = torch.tensor([0, S_img + n_valid, S_img + S_txt], dtype=torch.int32, device=dev)
    single = block_sparse_attention(qd, kd, vd, top_k, cu_seqlens_q=cu1, cu_seqlens_kv=cu1, text_blocks=tb,
                                    text_amp=amp, block_neighbor_list=nb_dev, shape_xfuse=True, p_remain_rates=p_rate)
    for r in range(N):
        want = torch.cat([single[:, r * S_loc:(r + 1) * S_loc], single[:, S_img:]], dim=1)
        assert torch.equal(results[r], want), f"rank {r} of {N}: simulated exchange differs from the single-rank op"

    # (b) the oracle's N-rank simulation (numpy), attention tolerance
    qn, kn, vn = to_np(q), to_np(k), to_np(v)
    shards = lambda t: [t[:, r * S_loc:(r + 1) * S_loc] for r in range(N)]
    sim = ou.simulate(shards(qn), shards(kn), shards(vn), qn[:, S_img:], kn[:, S_img:], vn[:, S_img:], top_k, n_valid,
                      "bfloat16", text_amp=amp, neighbors=nbm, p=p_rate)
    for r in range(N):
        err = np.abs(results[r].float().cpu().numpy() - sim[r])
        # (a borderline block may be selected differently by the numpy restatement of the pooling: allow a few rows)
        assert err.mean() <= 2e-3 and (err.max(-1) > 3e-2).mean() <= 0.02, (r, err.max(), err.mean())


@pytest.mark.parametrize("tag,grid,drop,amp,tb", [
    ("config2_full_stage", (32, 45, 80), 0.75, 0.0, 2),          # S_loc 14 400, top_k 8 * int(0.25 * 112) = 224
    ("config3_turbo_stage0", (32, 33, 60), 0.75, 0.431, 2),     # S_loc 7 920,  top_k 8 * int(0.25 * 61) = 120, text_amp 0.431
    ("config5_i2v_stage0", (32, 22, 40), 0.75, 1.016, 4),       # S_loc 3 520,  top_k 8 * int(0.25 * 27) = 48, four text blocks
])
def test_full_size_rank_share_of_eight_equals_single_rank_op(dev, tag, grid, drop, amp, tb):
    """One rank's REAL share of the 8-GPU configurations (BASELINE.json configs 2/3/5: 24 heads -> 3 heads x all 115 456 /
    63 616 keys per rank, S_loc = S_img / 8 not a multiple of 128, top_k = N * int((1 - r) * (S_loc // 128)),
    models_mul_block_gc_ha_multigpu.py:249-251; cu_seqlens rebuilt, xdit_ring_atten.py:105,183-184) through
    UlyssesAttenCarve with 8 simulated ranks, against the single-rank op on the same heads: bit for bit (VERDICT r5 weak 4:
    until now this ran at <= 3 200 tokens)."""
    from jenga_amd import gilbert as G
    from jenga_amd.modules import ulysses
    from jenga_amd.modules.attention import my_parallel_attention
    from jenga_amd.modules.attention_block_sparse import block_sparse_attention
    N, H = 8, 24
    t, h, w = grid
    S_img, S_txt = t * h * w, tb * 128
    nimg = S_img // 128
    S_loc = S_img // N
    assert S_img % (128 * 1) == 0 and S_img % N == 0 and S_loc % 128 != 0
    top_k = N * int((1 - drop) * (S_loc // 128))
    assert top_k == {"config2_full_stage": 224, "config3_turbo_stage0": 120, "config5_i2v_stage0": 48}[tag]
    n_valid, p_rate = 70, 0.3
    g = torch.Generator(device=dev).manual_seed(99)
    nb = nimg + tb
    # peaky block structure (block centroids) so that top_k, the p-rule and the neighbours all decide somewhere
    cent = torch.randn(1, nb, 1, H, 128, generator=g, device=dev) * 0.6
    mk = lambda c: (torch.randn(1, nb, 128, H, 128, generator=g, device=dev) + c).to(torch.bfloat16).view(1, nb * 128, H, 128)
    qd = mk(cent[:, torch.randint(0, nimg, (nb,), generator=g, device=dev)])
    kd = mk(cent)
    vd = torch.randn(1, nb * 128, H, 128, generator=g, device=dev).to(torch.bfloat16)
    nb_dev = G.gilbert_block_neighbor_mapping(t, h, w, 128).to(dev)
    assert nb_dev.shape == (nimg, nimg)

    world = SimWorld(N)
    results, errors = [None] * N, []

    def run(rank):
        try:
            torch.cuda.set_device(dev)
            sl = slice(rank * S_loc, (rank + 1) * S_loc)
            loc = lambda x: torch.cat([x[:, sl], x[:, S_img:]], dim=1)
            cu = torch.tensor([0, S_loc + n_valid, S_loc + S_txt], dtype=torch.int32, device=dev)
            sp = ulysses.UlyssesAttenCarve(exchange=SimExchange(world, rank))
            out = my_parallel_attention(sp, loc(qd), loc(kd), loc(vd), img_q_len=S_loc, img_kv_len=S_loc,
                                        cu_seqlens_q=cu, cu_seqlens_kv=cu, top_k=top_k, text_amp=amp,
                                        block_neighbor_list=nb_dev, p_remain_rates=p_rate)
            results[rank] = out.reshape(1, S_loc + S_txt, H, 128)
        except Exception as e:                                 # noqa: BLE001 - surfaced below
            errors.append((rank, repr(e)))
            world.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(N)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors
    torch.cuda.synchronize()
    cu1 = torch.tensor([0, S_img + n_valid, S_img + S_txt], dtype=torch.int32, device=dev)
    single = block_sparse_attention(qd, kd, vd, top_k, cu_seqlens_q=cu1, cu_seqlens_kv=cu1, text_blocks=tb,
                                    text_amp=amp, block_neighbor_list=nb_dev, shape_xfuse=True, p_remain_rates=p_rate)
    assert single.float().abs().max().item() > 0.05
    for r in range(N):
        want = torch.cat([single[:, r * S_loc:(r + 1) * S_loc], single[:, S_img:]], dim=1)
        assert torch.equal(results[r], want), f"{tag}: rank {r} of {N} differs from the single-rank op at full size"


@pytest.mark.parametrize("N", [2, 4])
def test_parallel_attention_is_the_dense_sequence_parallel_attention(dev, N):
    """attenion.py:198-251 (the reference's dense, non-Jenga sequence-parallel attention; its I2V blocks call it in their
    sequence-parallel branch): N simulated ranks against (a) the single-rank dense front-end `attention` on the valid rows --
    the same kernel with every block kept, so the image and valid-text rows must agree to the last bit where the row's
    block list is the same, and within the attention tolerance everywhere -- and (b) the oracle's dense restatement."""
    from jenga_amd.modules import ulysses
    from jenga_amd.modules.attention import attention, parallel_attention
    from oracle import attention as oa
    gen = torch.Generator().manual_seed(77 + N)
    H, nimg, tb = 8, 8, 2
    S_img, S_txt = nimg * 128, tb * 128
    S = S_img + S_txt
    q = torch.randn(1, S, H, 128, generator=gen).to(torch.bfloat16)
    k = torch.randn(1, S, H, 128, generator=gen).to(torch.bfloat16)
    v = torch.randn(1, S, H, 128, generator=gen).to(torch.bfloat16)
    S_loc, n_valid = S_img // N, 70
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    world = SimWorld(N)
    results, errors = [None] * N, []

    def run(rank):
        try:
            torch.cuda.set_device(dev)
            sl = slice(rank * S_loc, (rank + 1) * S_loc)
            loc = lambda t: torch.cat([t[:, sl], t[:, S_img:]], dim=1)
            cu = torch.tensor([0, S_loc + n_valid, S_loc + S_txt], dtype=torch.int32, device=dev)
            sp = ulysses.UlyssesAttenCarve(exchange=SimExchange(world, rank))
            out = parallel_attention(sp, loc(qd), loc(kd), loc(vd), img_q_len=S_loc, img_kv_len=S_loc, cu_seqlens_q=cu,
                                     cu_seqlens_kv=cu)
            results[rank] = out.reshape(1, S_loc + S_txt, H, 128)
        except Exception as e:                                 # noqa: BLE001 - surfaced below
            errors.append((rank, repr(e)))
            world.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    torch.cuda.synchronize()
    cu1 = torch.tensor([0, S_img + n_valid, S], dtype=torch.int32, device=dev)
    single = attention(qd, kd, vd, cu_seqlens_q=cu1, cu_seqlens_kv=cu1).reshape(1, S, H, 128)
    ref = oa.dense_varlen(to_np(q).transpose(0, 2, 1, 3), to_np(k).transpose(0, 2, 1, 3), to_np(v).transpose(0, 2, 1, 3),
                          cu1.cpu().numpy(), 128 ** -0.5, "bfloat16").transpose(0, 2, 1, 3)
    for r in range(N):
        got_img = results[r][:, :S_loc]
        got_txt = results[r][:, S_loc:S_loc + n_valid]
        got_pad = results[r][:, S_loc + n_valid:]            # the text-padding rows among themselves (attenion.py:222-247)
        want_img = single[:, r * S_loc:(r + 1) * S_loc]
        want_txt = single[:, S_img:S_img + n_valid]
        want_pad = single[:, S_img + n_valid:]
        assert got_pad.float().abs().max().item() > 0.05
        for got, want, rf in ((got_img, want_img, ref[:, r * S_loc:(r + 1) * S_loc]), (got_txt, want_txt, ref[:, S_img:S_img + n_valid]),
                              (got_pad, want_pad, ref[:, S_img + n_valid:])):
            e1 = (got.float() - want.float()).abs().max().item()
            e2 = np.abs(got.float().cpu().numpy() - rf)
            e3 = np.abs(want.float().cpu().numpy() - rf)
            assert e1 <= 1.6e-2, (r, "got-single", e1, "got-oracle", e2.max(), "single-oracle", e3.max())
            assert e2.max() <= 2e-2 and e2.mean() <= 1e-3, (r, e2.max(), e2.mean())
    with pytest.raises(TypeError):
        parallel_attention(object(), qd, kd, vd, S_img, S_img, cu1, cu1)
