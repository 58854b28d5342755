"""Diagnostics: N processes sharing one GPU, op level (UlyssesAttenCarve with the host-staged exchange of tests/helpers.py)
against the single-rank op.  python tools/diag_mp.py [N]"""
import json, os, socket, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as dist
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import inputs
    from helpers import HostStagedExchange
    from jenga_amd.modules import ulysses
    from jenga_amd.modules.attention import my_parallel_attention
    from jenga_amd.modules.attention_block_sparse import block_sparse_attention
    from oracle import gilbert as og
    N = world
    gen = torch.Generator().manual_seed(321 + N)
    H, nimg, tb = 8, 9, 2
    q, k = inputs.peaky_qk(gen, 1, H, nimg + tb, nimg + tb, 128, 0.8)
    q = q.transpose(1, 2).to(torch.bfloat16).contiguous(); k = k.transpose(1, 2).to(torch.bfloat16).contiguous()
    v = torch.randn(1, (nimg + tb) * 128, H, 128, generator=gen).to(torch.bfloat16)
    nbm = og.gilbert_block_neighbor_mapping(3, 12, 32, 128)
    S_img, S_txt = nimg * 128, tb * 128
    S_loc = S_img // N
    n_valid, amp, p_rate = 70, 0.25, 0.3
    top_k = N * int((1 - 0.5) * (S_loc // 128))
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    nb_dev = torch.from_numpy(nbm).to(dev)
    sl = slice(rank * S_loc, (rank + 1) * S_loc)
    loc = lambda t: torch.cat([t[:, sl], t[:, S_img:]], dim=1)
    cu = torch.tensor([0, S_loc + n_valid, S_loc + S_txt], dtype=torch.int32, device=dev)
    sp = ulysses.UlyssesAttenCarve(exchange=HostStagedExchange())
    out = my_parallel_attention(sp, loc(qd), loc(kd), loc(vd), img_q_len=S_loc, img_kv_len=S_loc, cu_seqlens_q=cu,
                                cu_seqlens_kv=cu, top_k=top_k, text_amp=amp, block_neighbor_list=nb_dev, p_remain_rates=p_rate)
    got = out.reshape(1, S_loc + S_txt, H, 128)
    cu1 = torch.tensor([0, S_img + n_valid, S_img + S_txt], dtype=torch.int32, device=dev)
    single = block_sparse_attention(qd, kd, vd, top_k, cu_seqlens_q=cu1, cu_seqlens_kv=cu1, text_blocks=tb, text_amp=amp,
                                    block_neighbor_list=nb_dev, shape_xfuse=True, p_remain_rates=p_rate)
    want = torch.cat([single[:, sl], single[:, S_img:]], dim=1)
    err = (got.float() - want.float()).abs()
    rec = {"rank": rank, "equal": bool(torch.equal(got, want)), "max": float(err.max()), "mean": float(err.mean()),
           "img_rows_bad": int((err[:, :S_loc].amax((-1, -2)) > 0).sum()), "txt_rows_bad": int((err[:, S_loc:].amax((-1, -2)) > 0).sum()),
           "heads_bad": [int(h) for h in torch.nonzero(err.amax((0, 1, 3)) > 0).flatten().tolist()]}
    json.dump(rec, open(os.path.join(outdir, f"r{rank}.json"), "w"))
    dist.destroy_process_group()


def worker_dit(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import copy
    import threading
    import torch.distributed as dist
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import HostStagedExchange, HostStagedGroup, SimExchange, SimGroup, SimWorld
    from jenga_amd import dit
    from jenga_amd.modules import ulysses
    from test_gpu_sp_dit import _model
    N = world
    latent, n_txt = (4, 16, 32), 256
    base = _model(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(1, 16, *latent, generator=g, device=dev, dtype=torch.bfloat16)
    text = torch.randn(1, n_txt, 64, generator=g, device=dev, dtype=torch.bfloat16)
    text2 = torch.randn(1, 32, generator=g, device=dev, dtype=torch.bfloat16)
    mask = torch.zeros(1, n_txt, dtype=torch.int64, device=dev); mask[:, :70] = 1
    gd = torch.tensor([6000.0], device=dev)

    def run(m):
        cos, sin = m.set_stage(latent, dev)
        m.sa_drop_rate, m.text_amp, m.p_remain_rates, m.enable_skip, m.num_steps = 0.5, 0.2, 0.3, True, 50
        m.cnt = 0
        o = m(x, torch.tensor([900.0], device=dev), text, mask, text2, cos, sin, gd, return_dict=False)
        torch.cuda.synchronize()
        return o

    orig = dit._select_top_k
    dit._select_top_k = lambda r, nblk: N * int((1 - r) * ((nblk * 128 // N) // 128))
    try:
        want = run(copy.deepcopy(base))
    finally:
        dit._select_top_k = orig
    rec = {"rank": rank}
    # (0) is the single-rank forward the same tensor in every process?  and the jenga_linear plans?
    from jenga_amd import _capi
    want2 = run(copy.deepcopy(base))          # without the top_k patch
    hw = want.float().cpu(); parts = [torch.empty_like(hw) for _ in range(world)]
    dist.all_gather(parts, hw)
    rec["want_equal_across_processes"] = bool(all(torch.equal(p_, parts[0]) for p_ in parts))
    rec["want_mean_diff_vs_rank0"] = float((hw - parts[0]).abs().mean())
    hw2 = want2.float().cpu(); parts2 = [torch.empty_like(hw2) for _ in range(world)]
    dist.all_gather(parts2, hw2)
    rec["want_unpatched_equal_across_processes"] = bool(all(torch.equal(p_, parts2[0]) for p_ in parts2))
    ch = _capi.linear_export_choices()
    ch = ch[torch.argsort(ch[:, :11].to(torch.float64) @ torch.arange(1, 12, dtype=torch.float64).pow(3))]
    chp = [torch.empty_like(ch) for _ in range(world)]
    try:
        dist.all_gather(chp, ch)
        rec["plans"] = int(ch.shape[0]); rec["plans_equal_across_processes"] = bool(all(torch.equal(c_, chp[0]) for c_ in chp))
        if not rec["plans_equal_across_processes"]:
            bad = (ch != chp[0]).any(1)
            rec["plans_differing"] = [[int(v) for v in row] for row in ch[bad][:4].tolist()] + [[int(v) for v in row] for row in chp[0][bad][:4].tolist()]
    except Exception as e:
        rec["plans_error"] = repr(e)
    # (1) thread-simulated ranks inside THIS process (dist is initialised here, unlike in tests/test_gpu_sp_dit.py)
    w = SimWorld(N); res = [None] * N; errs = []
    def th(r):
        try:
            torch.cuda.set_device(dev)
            ulysses.set_thread_sp_group(SimGroup(w, r))
            m = copy.deepcopy(base); ex = SimExchange(w, r)
            for blk in list(m.double_blocks) + list(m.single_blocks):
                blk.hybrid_seq_parallel_attn = ulysses.UlyssesAttenCarve(exchange=ex)
            res[r] = run(m)
        except Exception as e:
            errs.append(repr(e)); w.barrier.abort()
        finally:
            ulysses.set_thread_sp_group(None)
    ts = [threading.Thread(target=th, args=(r,)) for r in range(N)]
    [t.start() for t in ts]; [t.join() for t in ts]
    rec["sim_errors"] = errs
    if not errs:
        rec["sim_equal_want"] = bool(torch.equal(res[0], want))
    # (2) the real processes, host-staged exchange
    ulysses.init_sequence_parallel()
    ulysses.set_thread_sp_group(HostStagedGroup())
    m = copy.deepcopy(base); ex = HostStagedExchange()
    for blk in list(m.double_blocks) + list(m.single_blocks):
        blk.hybrid_seq_parallel_attn = ulysses.UlyssesAttenCarve(exchange=ex)
    got = run(m)
    err = (got.float() - want.float()).abs()
    rec.update({"mp_equal_want": bool(torch.equal(got, want)), "mp_max": float(err.max()), "mp_mean": float(err.mean()),
                "mp_frac_nonzero": float((err > 0).float().mean())})
    if not errs:
        rec["mp_equal_sim"] = bool(torch.equal(got, res[0]))
    json.dump(rec, open(os.path.join(outdir, f"r{rank}.json"), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "dit":
        worker = worker_dit
    import torch.multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(n, port, d), nprocs=n, join=True)
        for r in range(n):
            print(open(os.path.join(d, f"r{r}.json")).read())
