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


@pytest.mark.parametrize("H,nq_img,size", [(3, 72, 64), (2, 200, 7), (1, 130, 64)])
def test_cohort_start_barrier_is_bit_identical(dev, monkeypatch, H, nq_img, size):
    """JENGA_ATTN_COHORT (round-4 experiment): the workgroups of an XCD generation wait for each other before they start
    (arrival counters, bounded spin).  A scheduling device only: every query block is computed exactly once, outputs are
    bit-identical -- also when a generation is larger than what can be resident at once (the timeout lets it proceed)."""
    from jenga_amd import _capi
    if not _capi.has_experiments():
        pytest.skip("the cohort start barrier is part of the experiments library (JENGA_LIB=libjenga_amd_exp.so)")
    monkeypatch.setenv("JENGA_COHORT_SIZE", str(size))
    monkeypatch.setenv("JENGA_COHORT_TIMEOUT_US", "50")
    tb = 2
    q, k, v, mask = _rand_case(91 + nq_img, H, nq_img, tb, "bfloat16", 0.3, 0.0)
    nb = nq_img + tb
    idx, cnt = lists_from_mask(mask, dev)
    vt = _capi.pack_v(v.to(dev), nb)
    seqlens = torch.tensor([nq_img * 128 + 70], dtype=torch.int32, device=dev)
    run = lambda fl: _capi.bsattn_fwd(q.to(dev), k.to(dev), vt, seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3, nq_img, flags=fl)
    base = run(_capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED)
    for _ in range(2):       # twice: the counters are zeroed on the stream in front of every launch
        coh = run(_capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED | _capi.ATTN_COHORT)
        torch.cuda.synchronize()
        assert torch.equal(base, coh)
        both = run(_capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED | _capi.ATTN_COHORT | _capi.ATTN_BALANCE)
        torch.cuda.synchronize()       # the barrier on DRAWN blocks (generation = ticket / size; guests do not wait)
        assert torch.equal(base, both)


@pytest.mark.parametrize("period_us", [1, 37, 5000])
def test_rotated_list_walk_equals_the_ascending_walk_within_rounding(dev, monkeypatch, period_us):
    """JENGA_ATTN_ROTATE (round-4 experiment): every workgroup walks the unmasked part of its ascending list from a start
    rotated by the phase of a wall-clock cursor.  The set of (query block, kv block) pairs is unchanged; only the order of
    the online-softmax accumulation differs, so the result equals the ascending walk's within fp32 rounding of the running
    sums (the bound of the two-kernel comparison), and V == 1 still gives exactly-one rows."""
    from jenga_amd import _capi
    monkeypatch.setenv("JENGA_ROTATE_PERIOD_US", str(period_us))
    H, nq_img, tb = 3, 150, 2
    q, k, v, mask = _rand_case(1234, H, nq_img, tb, "bfloat16", 0.35, 0.0)
    nb = nq_img + tb
    idx, cnt = lists_from_mask(mask, dev)
    vt = _capi.pack_v(v.to(dev), nb)
    seqlens = torch.tensor([nq_img * 128 + 70], dtype=torch.int32, device=dev)
    base_fl = _capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED
    run = lambda fl, vt_: _capi.bsattn_fwd(q.to(dev), k.to(dev), vt_, seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3, nq_img, flags=fl)
    ref = run(base_fl, vt)
    rot = run(base_fl | _capi.ATTN_ROTATE, vt)
    torch.cuda.synchronize()
    d = (ref.float() - rot.float()).abs()
    assert float(d.max()) <= 2e-2 and float(d.mean()) <= 5e-4, (float(d.max()), float(d.mean()))
    ones = _capi.pack_v(torch.ones_like(v).to(dev), nb)
    o1 = run(base_fl | _capi.ATTN_ROTATE, ones)
    valid_rows = nq_img * 128 + 70
    assert torch.all((o1[:, :valid_rows].float() - 1).abs() <= 2 ** -7)


@pytest.mark.parametrize("slots", [64, 7])
def test_rotated_walk_position_mode_is_deterministic(dev, monkeypatch, slots):
    """JENGA_ROTATE_SLOTS (the deterministic form of JENGA_ATTN_ROTATE): the rotation of a workgroup is a pure function of
    its position in its XCD's launch queue -- two runs give the same bits; the result equals the ascending walk's within
    fp32 rounding of the running sums."""
    from jenga_amd import _capi
    if not _capi.has_experiments():
        pytest.skip("the position mode of the rotated walk is part of the experiments library")
    monkeypatch.setenv("JENGA_ROTATE_SLOTS", str(slots))
    H, nq_img, tb = 2, 200, 2
    q, k, v, mask = _rand_case(4321, H, nq_img, tb, "bfloat16", 0.3, 0.0)
    nb = nq_img + tb
    idx, cnt = lists_from_mask(mask, dev)
    vt = _capi.pack_v(v.to(dev), nb)
    seqlens = torch.tensor([nq_img * 128 + 70], dtype=torch.int32, device=dev)
    base_fl = _capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED
    run = lambda fl: _capi.bsattn_fwd(q.to(dev), k.to(dev), vt, seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3, nq_img, flags=fl)
    ref = run(base_fl)
    r1 = run(base_fl | _capi.ATTN_ROTATE)
    r2 = run(base_fl | _capi.ATTN_ROTATE)
    torch.cuda.synchronize()
    assert torch.equal(r1, r2)
    d = (ref.float() - r1.float()).abs()
    assert float(d.max()) <= 2e-2 and float(d.mean()) <= 5e-4, (float(d.max()), float(d.mean()))



@pytest.mark.parametrize("H,nq_img,extra", [(3, 72, "12"), (2, 200, "0"), (1, 130, "100"), (5, 67, "12")])
def test_balanced_launch_is_bit_identical(dev, monkeypatch, H, nq_img, extra):
    """JENGA_ATTN_BALANCE (round 4): a workgroup DRAWS its query block -- a ticket from the queue of the XCD it runs on,
    then from the fullest other queue -- on an oversubscribed grid.  Which workgroup computes a block does not enter the
    result: every block exactly once, outputs bit-identical to the static mapping; also without oversubscription, with
    twice the grid, with ragged last ranges (nq_img not a multiple of 8), and over more launches than there are counter
    sets (the sets are reused in turn behind an event)."""
    from jenga_amd import _capi
    monkeypatch.setenv("JENGA_BALANCE_EXTRA_PCT", extra)
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
    for i in range(70 if extra == "12" and H == 3 else 3):
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


def test_balanced_and_rotated_walk_together(dev, monkeypatch):
    """BALANCE | ROTATE: the drawn query block's list is walked from the clock cursor (the throughput mode's kernel); same
    pairs, another summation order -- the rotated walk's bound against the ascending one."""
    from jenga_amd import _capi
    monkeypatch.setenv("JENGA_ROTATE_PERIOD_US", "37")
    H, nq_img, tb = 3, 150, 2
    q, k, v, mask = _rand_case(4321, H, nq_img, tb, "bfloat16", 0.35, 0.0)
    nb = nq_img + tb
    idx, cnt = lists_from_mask(mask, dev)
    vt = _capi.pack_v(v.to(dev), nb)
    seqlens = torch.tensor([nq_img * 128 + 70], dtype=torch.int32, device=dev)
    base_fl = _capi.ATTN_XCD_REMAP | _capi.ATTN_LP | _capi.ATTN_SORTED
    run = lambda fl: _capi.bsattn_fwd(q.to(dev), k.to(dev), vt, seqlens, idx, cnt, nq_img, 128 ** -0.5, 0.3, nq_img, flags=fl)
    ref = run(base_fl)
    both = run(base_fl | _capi.ATTN_BALANCE | _capi.ATTN_ROTATE)
    torch.cuda.synchronize()
    d = (ref.float() - both.float()).abs()
    assert float(d.max()) <= 2e-2 and float(d.mean()) <= 5e-4, (float(d.max()), float(d.mean()))
    assert not torch.equal(ref, both) or float(d.max()) == 0.0


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
