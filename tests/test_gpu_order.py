"""GPU (-m gpu): the kept-count-aware launch order ("load imbalance -> work queue sorted by kept count";
the reference launches its grid in plain order, attention_block_triton_diffres.py:165).  jenga_order_by_count builds a
permutation per (batch, head): inside every segment of consecutive query blocks, descending kept count; jenga_bsattn_fwd
(LP kernel) maps launch position -> query block through it.  A scheduling hint only: outputs are bit-identical."""
import pytest
import torch

from test_gpu_pair import _rand_case, lists_from_mask

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("nq,seg", [(900, 113), (200, 25), (200, 200), (7, 3), (2048, 2048), (64, 8)])
def test_order_by_count_is_a_segmentwise_descending_permutation(dev, nq, seg):
    from jenga_amd import _capi
    g = torch.Generator(device=dev).manual_seed(nq + seg)
    cnt = torch.randint(1, 40, (2, 3, nq), generator=g, device=dev, dtype=torch.int32)
    order = _capi.order_by_count(cnt, seg)
    torch.cuda.synchronize()
    c, o = cnt.cpu(), order.cpu()
    for b in range(2):
        for h in range(3):
            for s0 in range(0, nq, seg):
                n = min(seg, nq - s0)
                blocks = o[b, h, s0:s0 + n]
                assert sorted(blocks.tolist()) == list(range(s0, s0 + n)), "not a permutation of the segment"
                key = [(-int(c[b, h, m]), int(m)) for m in blocks.tolist()]
                assert key == sorted(key), "not (descending count, ascending block)"


@pytest.mark.parametrize("flags", [9, 8])
def test_sorted_launch_order_is_bit_identical(dev, flags):
    from jenga_amd import _capi
    H, nq_img, tb = 3, 72, 2            # >= 64 query blocks: the XCD remap is on for flags 9 (segment = 9)
    q, k, v, mask = _rand_case(77, H, nq_img, tb, "bfloat16", 0.3, 0.0)
    g = torch.Generator().manual_seed(5)
    thin = torch.rand(1, H, nq_img, nq_img + tb, generator=g) < 0.3
    mask[:, :, ::2, :nq_img] &= thin[:, :, ::2, :nq_img]      # ragged kept counts
    for m in range(nq_img):
        mask[:, :, m, m] = True
    nb = nq_img + tb
    idx, cnt = lists_from_mask(mask, dev)
    vt = _capi.pack_v(v.to(dev), nb)
    seqlens = torch.tensor([nq_img * 128 + 70], dtype=torch.int32, device=dev)
    run = lambda fl, order=None: _capi.bsattn_fwd(q.to(dev), k.to(dev), vt, seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3,
                                                  nq_img, flags=fl, order=order)
    plain = run(flags)
    srt = run(flags | _capi.ATTN_SORTED)
    # an arbitrary permutation handed in by the caller works the same way
    perm = torch.stack([torch.randperm(nq_img, generator=g) for _ in range(H)]).view(1, H, nq_img).to(torch.int32).to(dev)
    arb = run(flags, order=perm)
    torch.cuda.synchronize()
    assert torch.equal(plain, srt) and torch.equal(plain, arb)
    with pytest.raises(ValueError):
        run(flags, order=perm[:, :, :-1].contiguous())


@pytest.mark.parametrize("H,nq_img", [(3, 72), (2, 200), (1, 130), (5, 67)])
def test_balanced_launch_is_bit_identical(dev, H, nq_img):
    """JENGA_ATTN_BALANCE (round 4): a workgroup DRAWS its query block -- a ticket from the queue of the XCD it runs on,
    then from the fullest other queue -- on an oversubscribed grid.  Which workgroup computes a block does not enter the
    result: every block exactly once, outputs bit-identical to the static mapping; also with ragged last ranges (nq_img
    not a multiple of 8) and over more launches than there are counter sets (the sets are reused in turn behind an event).
    (JENGA_BALANCE_EXTRA_PCT is read once per process since round 5: the oversubscription is the default 12 % here.)"""
    from jenga_amd import _capi
    tb = 2
    q, k, v, mask = _rand_case(191 + nq_img, H, nq_img, tb, "bfloat16", 0.3, 0.0)
    nb = nq_img + tb
    idx, cnt = lists_from_mask(mask, dev)
    vt = _capi.pack_v(v.to(dev), nb)
    seqlens = torch.tensor([nq_img * 128 + 70], dtype=torch.int32, device=dev)
    qd, kd = q.to(dev), k.to(dev)
    run = lambda fl, out=None: _capi.bsattn_fwd(qd, kd, vt, seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3, nq_img, flags=fl,
                                                out=out)
    base_fl = _capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED
    base = run(base_fl, torch.full((1, nb * 128, H, 128), 777.0, dtype=torch.bfloat16, device=dev))
    for i in range(70 if H == 3 else 3):
        out = torch.full_like(base, 777.0)      # a block nobody draws would keep the fill value
        run(base_fl | _capi.ATTN_BALANCE, out)
        if i % 23 == 0 or i < 3:
            torch.cuda.synchronize()
            assert torch.equal(base, out), i
    torch.cuda.synchronize()
    assert torch.equal(base, out)


def test_balanced_launches_from_concurrent_threads_and_streams(dev):
    """Ranks simulated by threads launch on one device at the same time, each on its own stream: every launch in flight has
    its own ticket counters, so no launch can draw from another's queue (a shared set would leave blocks uncomputed)."""
    import threading
    from jenga_amd import _capi
    H, nq_img, tb = 2, 136, 2
    nb = nq_img + tb
    base_fl = _capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED
    seqlens = torch.tensor([nq_img * 128 + 70], dtype=torch.int32, device=dev)
    cases = []
    for s in range(4):
        q, k, v, mask = _rand_case(700 + s, H, nq_img, tb, "bfloat16", 0.3, 0.0)
        idx, cnt = lists_from_mask(mask, dev)
        c = dict(q=q.to(dev), k=k.to(dev), vt=_capi.pack_v(v.to(dev), nb), idx=idx, cnt=cnt)
        c["want"] = _capi.bsattn_fwd(c["q"], c["k"], c["vt"], seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3, nq_img, flags=base_fl,
                                     out=torch.full((1, nb * 128, H, 128), 777.0, dtype=torch.bfloat16, device=dev))
        cases.append(c)
    torch.cuda.synchronize()
    errors = []

    def worker(c):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for i in range(40):
                    out = torch.full_like(c["want"], 777.0)
                    _capi.bsattn_fwd(c["q"], c["k"], c["vt"], seqlens, c["idx"], c["cnt"], nq_img, 128 ** -0.5, 0.3, nq_img,
                                     flags=base_fl | _capi.ATTN_BALANCE, out=out)
                    if i % 13 == 0:
                        st.synchronize()
                        if not torch.equal(out, c["want"]):
                            errors.append(i)
                st.synchronize()
                if not torch.equal(out, c["want"]):
                    errors.append(-1)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(c,)) for c in cases]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_balanced_launch_under_stream_capture_falls_back_to_the_static_mapping(dev):
    """A capturing stream gets no ticket counters (an event recorded inside a capture cannot order the set against launches
    outside it): the launch inside a HIP graph runs the static mapping, the graph replays, and every replay equals the
    eager (balanced) result bit for bit."""
    from jenga_amd import _capi
    H, nq_img, tb = 2, 136, 2
    nb = nq_img + tb
    q, k, v, mask = _rand_case(77, H, nq_img, tb, "bfloat16", 0.3, 0.0)
    idx, cnt = lists_from_mask(mask, dev)
    qd, kd = q.to(dev), k.to(dev)
    vt = _capi.pack_v(v.to(dev), nb)
    seqlens = torch.tensor([nq_img * 128 + 70], dtype=torch.int32, device=dev)
    fl = _capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED | _capi.ATTN_BALANCE
    order = _capi.order_by_count(cnt, (nq_img + 7) // 8)
    want = _capi.bsattn_fwd(qd, kd, vt, seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3, nq_img, flags=fl, order=order)
    out = torch.zeros_like(want)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            _capi.bsattn_fwd(qd, kd, vt, seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3, nq_img, flags=fl, order=order, out=out)
    for _ in range(3):
        out.fill_(777.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


@pytest.mark.parametrize("kernel", ["lp", "pair"])
def test_full_size_launch_writes_every_query_block_exactly_once(dev, kernel):
    """BASELINE.json configs[1] at full size (900 image + 2 text query blocks x 24 heads, sa-drop 0.8): the output buffer is
    pre-filled with a sentinel; after ONE balanced launch (a) no row of any of the 902 x 24 query blocks still holds it -- a
    workgroup that gave up on its draw would leave one --, (b) with V == 1 every row valid for the reference is exactly 1.0
    within an ulp (softmax weights sum to one: a block computed partially or against a wrong list would not be), rows behind the kv length are exactly 0, (c) the result equals the static mapping's
    bit for bit.  The ordering of the ticket atomics rests on data dependences the compiler cannot see (csrc/lp_balance.h):
    this is their runtime check."""
    from jenga_amd import _capi
    H, nq_img, tb = 24, 900, 2
    nb = nq_img + tb
    S = nb * 128
    g = torch.Generator(device=dev).manual_seed(11)
    q = torch.randn(1, S, H, 128, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    k = torch.randn(1, S, H, 128, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    v = torch.ones(1, S, H, 128, device=dev, dtype=torch.bfloat16)
    top_k = int((1 - 0.8) * nq_img)
    keep = torch.rand(1, H, nq_img, nb, generator=g, device=dev)
    thr = keep[..., :nq_img].kthvalue(top_k, dim=-1, keepdim=True).values
    mask = keep <= thr
    mask[..., nq_img:] = True
    ar = torch.arange(nq_img, device=dev)
    mask[:, :, ar, ar] = True
    order = torch.argsort((~mask).to(torch.int8), dim=-1, stable=True).to(torch.int32).contiguous()
    cnt = mask.sum(-1).to(torch.int32).contiguous()
    vt = _capi.pack_v(v, nb)
    seqlen = nq_img * 128 + 64
    seqlens = torch.tensor([seqlen], dtype=torch.int32, device=dev)
    SENT = 777.0
    fl = _capi.ATTN_LP_FLAGS if kernel == "lp" else _capi.ATTN_PAIR_FLAGS
    out = torch.full((1, S, H, 128), SENT, dtype=torch.bfloat16, device=dev)
    _capi.bsattn_fwd(q, k, vt, seqlens, order, cnt, nq_img, 128 ** -0.5, 0.0, nq_img, flags=fl, out=out)
    static = torch.full_like(out, SENT)
    _capi.bsattn_fwd(q, k, vt, seqlens, order, cnt, nq_img, 128 ** -0.5, 0.0, nq_img, flags=fl & ~_capi.ATTN_BALANCE,
                     out=static)
    torch.cuda.synchronize()
    assert not bool((out == SENT).any()), "a query block was never written"
    img_valid = out[:, :seqlen]
    # (P is rounded to bf16 before P.V, l sums the unrounded values: 1 within an ulp of the bf16 output)
    assert float((img_valid.float() - 1).abs().max()) <= 2.0 ** -7, float((img_valid.float() - 1).abs().max())
    assert bool((out[:, seqlen:nq_img * 128] == 0).all()) if seqlen < nq_img * 128 else True
    # text rows (>= nq_img * 128) see every key without a length mask: 1.0 as well; image rows >= seqlen do not exist here
    assert float((out[:, nq_img * 128:].float() - 1).abs().max()) <= 2.0 ** -7
    assert torch.equal(out, static)
