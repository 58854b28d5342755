"""Diagnostics: which op of the single-rank forward is not bit-stable when ANOTHER process shares the GPU?
python tools/diag_det.py   (spawns 2 processes on device 0; each repeats every op R times and counts distinct results)"""
import json, os, socket, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import copy
    import torch.distributed as dist
    import torch.nn.functional as F
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jenga_amd import _capi
    from test_gpu_sp_dit import _model
    R = int(os.environ.get('DIAG_R', '30'))
    rec = {"rank": rank}

    def distinct(fn):
        seen = []
        for _ in range(R):
            o = fn(); torch.cuda.synchronize()
            if not any(torch.equal(o, s_) for s_ in seen):
                seen.append(o.clone())
        return len(seen)

    base = _model(dev)
    C = base.hidden_size
    g = torch.Generator(device=dev).manual_seed(7)
    dist.barrier()
    for M in (() if os.environ.get("DIAG_ONLY_FORWARD") else (512, 256, 768)):
        x = torch.randn(1, M, C, generator=g, device=dev, dtype=torch.bfloat16)
        for name, N_ in (("qkv", 3 * C), ("proj", C), ("fc1", 4 * C)):
            w = torch.randn(N_, C, generator=g, device=dev, dtype=torch.bfloat16) * 0.05
            b = torch.randn(N_, generator=g, device=dev, dtype=torch.bfloat16)
            rec[f"F.linear M{M} {name}"] = distinct(lambda: F.linear(x, w, b))
            rec[f"jenga_linear M{M} {name}"] = distinct(lambda: _capi.linear(x, w, b))
        w2 = torch.randn(C, 4 * C, generator=g, device=dev, dtype=torch.bfloat16) * 0.05
        x2 = torch.randn(1, M, 4 * C, generator=g, device=dev, dtype=torch.bfloat16)
        rec[f"F.linear M{M} fc2"] = distinct(lambda: F.linear(x2, w2))
    # the epilogue GEMMs and the custom kernels of one block at the model's shapes (S = 512 + 256, 8 heads)
    from jenga_amd.modules import attention_block_sparse as op
    from oracle import gilbert as og
    for M in (() if os.environ.get("DIAG_ONLY_FORWARD") else (512, 256, 768)):
        x = torch.randn(1, M, C, generator=g, device=dev, dtype=torch.bfloat16)
        w = torch.randn(C, C, generator=g, device=dev, dtype=torch.bfloat16) * 0.05
        w4 = torch.randn(4 * C, C, generator=g, device=dev, dtype=torch.bfloat16) * 0.05
        w4b = torch.randn(C, 4 * C, generator=g, device=dev, dtype=torch.bfloat16) * 0.05
        x4 = torch.randn(1, M, 4 * C, generator=g, device=dev, dtype=torch.bfloat16)
        b = torch.randn(C, generator=g, device=dev, dtype=torch.bfloat16)
        b4 = torch.randn(4 * C, generator=g, device=dev, dtype=torch.bfloat16)
        gate = torch.randn(C, generator=g, device=dev, dtype=torch.bfloat16)
        res = torch.randn(1, M, C, generator=g, device=dev, dtype=torch.bfloat16)
        rec[f"jenga_linear gate+res M{M}"] = distinct(lambda: _capi.linear(x, w, b.float() * gate.float(), gate=gate, res=res))
        rec[f"jenga_linear gelu M{M}"] = distinct(lambda: _capi.linear(x, w4, b4, act=_capi.ACT_GELU_TANH))
        rec[f"jenga_linear fc2 gate+res M{M}"] = distinct(lambda: _capi.linear(x4, w4b, b.float() * gate.float(), gate=gate, res=res))
    H_, S_, tb = 8, 768, 2
    q = torch.randn(1, S_, H_, 128, generator=g, device=dev, dtype=torch.bfloat16)
    k = torch.randn(1, S_, H_, 128, generator=g, device=dev, dtype=torch.bfloat16)
    v = torch.randn(1, S_, H_, 128, generator=g, device=dev, dtype=torch.bfloat16)
    nbm = torch.from_numpy(og.gilbert_block_neighbor_mapping(2, 8, 16, 128)).to(dev)   # 512 tokens = 4 blocks
    cu = torch.tensor([0, 512 + 70, S_], dtype=torch.int32, device=dev)
    rec["block_sparse_attention op"] = distinct(lambda: op.block_sparse_attention(q, k, v, 2, cu_seqlens_q=cu, cu_seqlens_kv=cu,
                                               text_blocks=tb, text_amp=0.2, block_neighbor_list=nbm, p_remain_rates=0.3))
    rec["block_pool"] = distinct(lambda: _capi.block_pool(k, 6))
    qp, kp = _capi.block_pool(q, 4), _capi.block_pool(k, 6)
    rec["block_select idx"] = distinct(lambda: _capi.block_select(qp, kp, nbm, 4, tb, 2, 0.3)[1])
    rec["pack_v"] = distinct(lambda: _capi.pack_v(v, 6))
    for M in (512, 256, 768):
        xx = torch.randn(1, M, C, generator=g, device=dev, dtype=torch.bfloat16)
        sh_ = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
        sc_ = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
        rec[f"ln_modulate M{M}"] = distinct(lambda: _capi.ln_modulate(xx, sh_, sc_))
        yy = torch.randn(1, M, C, generator=g, device=dev, dtype=torch.bfloat16)
        rec[f"gate_residual M{M}"] = distinct(lambda: _capi.gate_residual(xx, yy, sh_))
        ix = torch.randperm(M, generator=torch.Generator().manual_seed(3)).to(dev)
        rec[f"gather_rows M{M}"] = distinct(lambda: _capi.gather_rows(xx, ix))
    # the whole forward
    latent, n_txt = (4, 16, 32), 256
    x = torch.randn(1, 16, *latent, generator=g, device=dev, dtype=torch.bfloat16)
    text = torch.randn(1, n_txt, 64, generator=g, device=dev, dtype=torch.bfloat16)
    text2 = torch.randn(1, 32, generator=g, device=dev, dtype=torch.bfloat16)
    mask = torch.zeros(1, n_txt, dtype=torch.int64, device=dev); mask[:, :70] = 1
    gd = torch.tensor([6000.0], device=dev)
    m = copy.deepcopy(base)
    cos, sin = m.set_stage(latent, dev)
    m.sa_drop_rate, m.text_amp, m.p_remain_rates, m.enable_skip, m.num_steps = 0.5, 0.2, 0.3, True, 50

    def fwd():
        m.cnt = 0
        return m(x, torch.tensor([900.0], device=dev), text, mask, text2, cos, sin, gd, return_dict=False)
    rec["forward"] = distinct(fwd)
    # which module's output varies first?
    order, hashes = [], {}

    def h(t):
        t = t.contiguous()
        if t.dtype in (torch.bfloat16, torch.float16):
            return int(t.view(torch.int16).to(torch.int64).sum().item())
        if t.dtype == torch.float32:
            return int(t.view(torch.int32).to(torch.int64).sum().item())
        return int(t.to(torch.int64).sum().item())

    def hook(name):
        def f(mod, inp, out):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            v = tuple(h(o) for o in outs if torch.is_tensor(o))
            if name not in hashes:
                order.append(name); hashes[name] = set()
            hashes[name].add(v)
        return f
    hs = [mod.register_forward_hook(hook(n)) for n, mod in m.named_modules() if n]
    for _ in range(12):
        fwd(); torch.cuda.synchronize()
    for x_ in hs:
        x_.remove()
    rec["varying_modules_in_order"] = [(n, len(hashes[n])) for n in order if len(hashes[n]) > 1][:12]
    rec["stable_before_first"] = [n for n in order[: order.index(rec["varying_modules_in_order"][0][0])]][-6:] if rec["varying_modules_in_order"] else "all stable"
    # every _capi call and F.linear of the forward: the first call whose OUTPUT varies although its INPUTS did not
    import types
    calls = {}          # call index -> (name, set(in hashes), set(out hashes))
    ctr = [0]

    def tens(o):
        if torch.is_tensor(o):
            return [o]
        if isinstance(o, (tuple, list)):
            return [t for x_ in o for t in tens(x_)]
        return []

    def wrap(name, fn):
        def f(*a, **kw):
            hin = tuple(h(t) for t in tens(list(a) + list(kw.values())) if t.is_cuda)
            out = fn(*a, **kw)
            torch.cuda.synchronize()
            hout = tuple(h(t) for t in tens(out) if t.is_cuda)
            i = ctr[0]; ctr[0] += 1
            c = calls.setdefault(i, (name, set(), set()))
            c[1].add(hin); c[2].add(hout)
            if name in ("ln_modulate", "qk_norm_rope_pool"):
                outs = [t for t in tens(out) if t.is_cuda] or [t for t in tens(list(a) + list(kw.values())) if t.is_cuda]
                key = (i, hin)
                if key not in first_out:
                    first_out[key] = [t.clone() for t in outs]
                else:
                    for ti, (t0, t1) in enumerate(zip(first_out[key], outs)):
                        if t0.shape == t1.shape and not torch.equal(t0, t1) and len(diffs) < 6:
                            d = (t0.float() - t1.float()).abs()
                            flat = d.reshape(-1, d.shape[-1]) if d.dim() > 1 else d.reshape(1, -1)
                            rows = torch.nonzero(flat.amax(-1) > 0).flatten()
                            diffs.append({"call": i, "name": name, "tensor": ti, "shape": list(t0.shape), "n_diff": int((d > 0).sum()),
                                          "rows_first": rows[:6].tolist(), "n_rows": int(rows.numel()), "max": float(d.max()),
                                          "cols_first": torch.nonzero(flat[rows[0]] > 0).flatten()[:8].tolist()})
            return out
        return f
    first_out, diffs = {}, []
    saved = {}
    for n_ in ("gather_rows", "rmsnorm_rope", "qk_norm_rope_pool", "ln_modulate", "gate_residual", "gelu_tanh", "linear", "block_pool",
               "block_select", "pack_v", "bsattn_fwd"):
        saved[n_] = getattr(_capi, n_); setattr(_capi, n_, wrap(n_, saved[n_]))
    flin = F.linear
    F.linear = wrap("F.linear", flin)
    torch.nn.functional.linear = F.linear
    for _ in range(12):
        ctr[0] = 0
        fwd(); torch.cuda.synchronize()
    for n_, f_ in saved.items():
        setattr(_capi, n_, f_)
    F.linear = flin
    bad = [(i, c[0], len(c[1]), len(c[2])) for i, c in sorted(calls.items()) if len(c[2]) > 1]
    rec["calls_total"] = len(calls)
    rec["first_varying_calls(idx,name,n_in,n_out)"] = bad[:8]
    rec["origin(inputs stable, output varies)"] = [b_ for b_ in bad if b_[2] == 1][:8]
    rec["diffs"] = diffs
    rec = {k: v for k, v in rec.items() if not isinstance(v, int) or k in ("rank", "forward", "calls_total") or v != 1}
    json.dump(rec, open(os.path.join(outdir, f"r{rank}.json"), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    n = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(n, port, d), nprocs=n, join=True)
        for r in range(n):
            print(open(os.path.join(d, f"r{r}.json")).read())
