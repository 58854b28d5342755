import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jenga_amd import _capi
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for name, M, K, N in (("linear2", 115456, 15360, 3072), ("fc2", 115200, 12288, 3072), ("proj", 115200, 3072, 3072)):
    x = torch.randn(1, M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    gate = torch.randn(1, N, generator=g, device=dev).to(torch.bfloat16)
    r = torch.randn(1, M, N, generator=g, device=dev).to(torch.bfloat16)
    o = torch.empty_like(r)
    lin = torch.nn.Linear(K, N, dtype=torch.bfloat16, device=dev)
    variants = {"torch_linear": lambda: lin(x), "plain_bias": lambda: _capi.linear(x, w, b, out=o),
                "bias_res": lambda: _capi.linear(x, w, b, res=r, out=o), "bias_gate": lambda: _capi.linear(x, w, b, gate=gate, out=o),
                "bias_gate_res": lambda: _capi.linear(x, w, b, gate=gate, res=r, out=o),
                "bias_res_inplace": lambda: _capi.linear(x, w, b, res=o, out=o)}
    for vn, fn in variants.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        out[f"{name}.{vn}"] = round(e0.elapsed_time(e1) / 30, 3)
    del x, w, r, o, lin
print(json.dumps(out))
