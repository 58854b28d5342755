"""Diagnostics: process 0 runs the DiT forward in a loop (the load); process 1 repeats simple ops -- torch's own and this
library's -- on fixed inputs and counts how often a result differs from the first one.  Two processes on device 0."""
import json, os, socket, sys, tempfile, time
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
    g = torch.Generator(device=dev).manual_seed(7)
    secs = float(os.environ.get("DIAG_SECS", "20"))
    load = os.environ.get("DIAG_LOAD", "forward")
    rec = {"rank": rank, "load": load}
    dist.barrier()
    if rank == 0:
        n = 0
        t0 = time.time()
        if load == "forward":
            base = _model(dev)
            latent, n_txt = (4, 16, 32), 256
            x = torch.randn(1, 16, *latent, generator=g, device=dev, dtype=torch.bfloat16)
            text = torch.randn(1, n_txt, 64, generator=g, device=dev, dtype=torch.bfloat16)
            text2 = torch.randn(1, 32, generator=g, device=dev, dtype=torch.bfloat16)
            mask = torch.zeros(1, n_txt, dtype=torch.int64, device=dev); mask[:, :70] = 1
            gd = torch.tensor([6000.0], device=dev)
            m = copy.deepcopy(base)
            cos, sin = m.set_stage(latent, dev)
            m.sa_drop_rate, m.text_amp, m.p_remain_rates, m.enable_skip, m.num_steps = 0.5, 0.2, 0.3, True, 50
            while time.time() - t0 < secs:
                m.cnt = 0
                m(x, torch.tensor([900.0], device=dev), text, mask, text2, cos, sin, gd, return_dict=False)
                torch.cuda.synchronize(); n += 1
        elif load == "torch":       # torch-only load: GEMMs + elementwise
            a = torch.randn(2048, 2048, generator=g, device=dev, dtype=torch.bfloat16)
            while time.time() - t0 < secs:
                b = (a @ a).tanh_(); b = F.layer_norm(b, (2048,)); torch.cuda.synchronize(); n += 1
        else:
            while time.time() - t0 < secs:
                time.sleep(0.01)
        rec["load_iterations"] = n
    else:
        C = 1024
        xx = torch.randn(1, 512, C, generator=g, device=dev, dtype=torch.bfloat16)
        sh_ = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
        sc_ = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
        w = torch.randn(C, C, generator=g, device=dev, dtype=torch.bfloat16) * 0.05
        ops = {
            "jenga ln_modulate": lambda: _capi.ln_modulate(xx, sh_, sc_),
            "jenga gate_residual": lambda: _capi.gate_residual(xx, xx, sh_),
            "torch layer_norm*scale+shift": lambda: F.layer_norm(xx, (C,)) * (1 + sc_) + sh_,
            "torch add": lambda: xx + xx,
            "torch F.linear": lambda: F.linear(xx, w),
            "torch softmax": lambda: torch.softmax(xx.float(), -1),
        }
        from oracle import gilbert as og
        H_, S_, tb = 8, 1280 + 256, 2
        q = torch.randn(1, S_, H_, 128, generator=g, device=dev, dtype=torch.bfloat16)
        k = torch.randn(1, S_, H_, 128, generator=g, device=dev, dtype=torch.bfloat16)
        v = torch.randn(1, S_, H_, 128, generator=g, device=dev, dtype=torch.bfloat16)
        nb = S_ // 128
        qp, kp = _capi.block_pool(q, nb - tb), _capi.block_pool(k, nb)
        vt = _capi.pack_v(v, nb)
        _, idx, cnt = _capi.block_select(qp, kp, None, nb - tb, tb, 3, 0.5)
        seqlens = torch.tensor([S_ - 100], dtype=torch.int32, device=dev)
        wq = torch.randn(128, generator=g, device=dev, dtype=torch.bfloat16)
        cos = torch.randn(S_ - 256, 128, generator=g, device=dev); sin = torch.randn(S_ - 256, 128, generator=g, device=dev)
        ops.update({
            "jenga block_select mask": lambda: _capi.block_select(qp, kp, None, nb - tb, tb, 3, 0.5, want_mask=True, want_lists=False)[0],
            "jenga block_pool": lambda: _capi.block_pool(k, nb),
            "jenga pack_v": lambda: _capi.pack_v(v, nb),
            "jenga rmsnorm_rope": lambda: _capi.rmsnorm_rope(q, wq, cos, sin),
            "jenga bsattn LP (29)": lambda: _capi.bsattn_fwd(q, k, vt, seqlens, idx, cnt, nb - tb, 128 ** -0.5, 0.2, nb - tb, flags=29),
            "jenga bsattn pair (85)": lambda: _capi.bsattn_fwd(q, k, vt, seqlens, idx, cnt, nb - tb, 128 ** -0.5, 0.2, nb - tb, flags=85),
            "jenga bsattn round-1 (1)": lambda: _capi.bsattn_fwd(q, k, vt, seqlens, idx, cnt, nb - tb, 128 ** -0.5, 0.2, nb - tb, flags=1),
            "jenga gather_rows": lambda: _capi.gather_rows(xx, torch.arange(511, -1, -1, device=dev)),
            "jenga linear gelu": lambda: _capi.linear(xx, w, None, act=_capi.ACT_GELU_TANH),
        })
        # the rest of the C ABI's device kernels
        oq, ok_ = torch.empty_like(q), torch.empty_like(k)
        qpo = torch.zeros(1, H_, nb, 128, device=dev, dtype=torch.bfloat16); kpo = torch.zeros_like(qpo)
        cosf = torch.randn(S_, 128, generator=g, device=dev); sinf = torch.randn(S_, 128, generator=g, device=dev)

        def fused():
            _capi.qk_norm_rope_pool(q, k, wq, wq, cosf, sinf, oq, ok_, qpool=qpo, kpool=kpo)
            return torch.cat([oq.flatten(), ok_.flatten(), qpo.flatten(), kpo.flatten()])
        xf = torch.randn(1, 512, 1536, generator=g, device=dev)
        wf = torch.randn(1536, generator=g, device=dev); bf = torch.randn(1536, generator=g, device=dev)
        yb = torch.randn(1, 512, 1536, generator=g, device=dev, dtype=torch.bfloat16)
        c64 = torch.randn(S_, 64, generator=g, device=dev, dtype=torch.float64); s64 = torch.randn(S_, 64, generator=g, device=dev, dtype=torch.float64)
        xw = torch.randn(1, S_, 12 * 128, generator=g, device=dev, dtype=torch.bfloat16)
        ww = torch.randn(12 * 128, generator=g, device=dev)
        ops.update({
            "jenga qk_norm_rope_pool": fused,
            "jenga gelu_tanh": lambda: _capi.gelu_tanh(xx),
            "jenga wan_ln_modulate": lambda: _capi.wan_ln_modulate(xf, wf, bf, wf, bf),
            "jenga wan_gate_residual": lambda: _capi.wan_gate_residual(xf, yb, wf),
            "jenga rmsnorm_rows": lambda: _capi.rmsnorm_rows(xw, ww, 1e-6),
            "jenga wan_norm_rope": lambda: _capi.wan_norm_rope(xw, ww, c64, s64, S_, 1e-6),
            "jenga rope_complex": lambda: _capi.rope_complex(q, c64, s64, S_),
            "jenga cross_attn": lambda: _capi.cross_attn_fwd(q[:, :512], k, v),
            "jenga ulysses pack": lambda: _capi.ulysses_pack_heads(q, 4),
            "jenga ulysses unpack": lambda: _capi.ulysses_unpack_heads(_capi.ulysses_pack_heads(q, 4), 4),
            "jenga linear gate+res": lambda: _capi.linear(xx, w, sh_.float().reshape(-1), gate=sc_.reshape(-1), res=xx),
        })
        first = {nm: f() for nm, f in ops.items()}
        torch.cuda.synchronize()
        bad = {nm: 0 for nm in ops}; runs = 0
        t0 = time.time()
        while time.time() - t0 < secs:
            for nm, f in ops.items():
                o = f()
                if not torch.equal(o, first[nm]):
                    bad[nm] += 1
                    if nm == "jenga ln_modulate" and bad[nm] <= 3:
                        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                        torch.save({"x": xx.cpu(), "shift": sh_.cpu(), "scale": sc_.cpu(), "good": first[nm].cpu(), "bad": o.cpu()},
                                   os.path.join(ROOT, "gpurun_out", f"diag_ln_{bad[nm]}.pt"))
            runs += 1
        rec["runs_per_op"] = runs; rec["mismatches"] = bad
    dist.barrier()
    json.dump(rec, open(os.path.join(outdir, f"r{rank}.json"), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(2, port, d), nprocs=2, join=True)
        for r in range(2):
            print(open(os.path.join(d, f"r{r}.json")).read())
