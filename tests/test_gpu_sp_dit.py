"""GPU (-m gpu): N-rank parity of the WHOLE sequence-parallel DiT forward.

Reference: jenga_hyvideo_multigpu.py:109-201 (new_forward: chunk :174, RoPE chunk :176-177, all-gather :193, scatter
:195) and :207-331 (transformer_sub_forward: the per-step skip decision :233-236, the residual cache), with
models_mul_block_gc_ha_multigpu.py:249-251 (top_k = N * int((1 - r) * local_blocks)).

N simulated ranks = N threads of one process on one GPU, each with ITS OWN copy of the model (own step counter, own
residual cache), its own sequence-parallel group (tests/helpers.SimGroup: all_gather really concatenates the ranks'
shards) and its own exchange (SimExchange: chunk r of rank p's send buffer really becomes chunk p of rank r's receive
buffer).  They run JengaHYVideoDiT.forward over computed -> skipped -> computed steps with enable_skip on.  Every
rank's output must equal the single-rank forward run with the multi-GPU top_k rule.

Bit-exactness: the hot path (gather, norm / RoPE, selection, attention, exchanges, scatter) is bit-exact by
construction; the token-wise GEMMs run on S_img / N rows instead of S_img rows, and hipBLASLt may pick another kernel
(another fp32 summation order) for another M.  The test therefore asserts torch.equal where the GEMMs cooperate and
otherwise a bound of two bf16 ulps of the value + 0.02 on >= 99.5 % of the elements (a borderline block may be
selected differently after a one-ulp change of a pooled score), and records which of the two happened in
gpurun_out/parity_records/sp_dit.json."""
import copy
import json
import os
import threading

import pytest
import torch

from helpers import SimExchange, SimGroup, SimWorld

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _model(dev, seed=0):
    from jenga_amd import dit
    dit.HUNYUAN_VIDEO_CONFIG["tiny8"] = dict(mm_double_blocks_depth=2, mm_single_blocks_depth=2,
                                             rope_dim_list=[16, 56, 56], hidden_size=1024, heads_num=8,
                                             mlp_width_ratio=4, guidance_embed=True)
    m = dit.JengaHYVideoDiT(config="tiny8", text_states_dim=64, text_states_dim_2=32, dtype=torch.bfloat16, device=dev)
    return m.init_synthetic_weights(0.03, seed=seed)


def _record(name, rec):
    d = os.path.join(ROOT, "gpurun_out", "parity_records")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "sp_dit.json")
        allrec = json.load(open(path)) if os.path.exists(path) else {}
        allrec[name] = rec
        json.dump(allrec, open(path, "w"), indent=1)
    except OSError:
        pass


class _ReferenceSignatureOnly(torch.nn.Module):
    """A sequence-parallel module that only offers xFuserLongContextAttention's forward (no forward_qkv): the blocks
    then take the unfused path (norm / RoPE kernels, then the reference-signature call)."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, *a, **kw):
        return self.inner(*a, **kw)


# (N, latent T H W, text tokens, i2v, fused prologue): image tokens = T * H/2 * W/2
CASES = [
    (2, (4, 16, 32), 256, False, True),     # 512 tokens: S_loc = 256 (a multiple of 128)
    (8, (4, 40, 80), 256, False, True),     # 3200 tokens = 25 blocks: S_loc = 400 (NOT a multiple of 128), top_k 8 vs 12
    (4, (3, 24, 64), 512, True, True),      # I2V token_replace: 1152 tokens = 9 blocks, S_loc = 288, 4 text blocks
    (8, (4, 40, 80), 256, False, False),    # the reference-signature path (separate norm / RoPE / pack kernels)
    (2, (3, 24, 64), 512, True, False),
    # round 5, JENGA_ULYSSES_PIPELINE: the rank's H/N heads exchanged and attended one head at a time (4 / 2 head groups)
    (2, (4, 16, 32), 256, False, True, True),
    (4, (3, 24, 64), 512, True, True, True),
    (2, (4, 40, 80), 256, False, True, True),
]
CASES = [c if len(c) == 6 else c + (False,) for c in CASES]


@pytest.mark.parametrize("N,latent,n_txt,i2v,fused,pipe", CASES)
def test_sp_forward_n_ranks_equals_single_rank(dev, N, latent, n_txt, i2v, fused, pipe):
    from jenga_amd import dit
    from jenga_amd.modules import ulysses
    base = _model(dev)
    g = torch.Generator(device=dev).manual_seed(11 + N)
    x = torch.randn(1, 16, *latent, generator=g, device=dev, dtype=torch.bfloat16)
    text = torch.randn(1, n_txt, 64, generator=g, device=dev, dtype=torch.bfloat16)
    text2 = torch.randn(1, 32, generator=g, device=dev, dtype=torch.bfloat16)
    mask = torch.zeros(1, n_txt, dtype=torch.int64, device=dev)
    mask[:, :70] = 1
    gd = torch.tensor([6000.0], device=dev)
    drop, amp, p_rate = 0.5, 0.2, 0.3
    steps = [(0, 900.0), (5, 700.0), (7, 500.0)]     # NON_SKIP_STEPS has 0 and 7, not 5: computed, skipped, computed
    assert 0 in dit.NON_SKIP_STEPS and 7 in dit.NON_SKIP_STEPS and 5 not in dit.NON_SKIP_STEPS
    S_img = latent[0] * (latent[1] // 2) * (latent[2] // 2)
    assert S_img % N == 0 and S_img % 128 == 0

    def configure(m):
        cos, sin = m.set_stage(latent, dev)
        m.sa_drop_rate, m.text_amp, m.p_remain_rates, m.enable_skip = drop, amp, p_rate, True
        m.num_steps = 50
        m.i2v_condition_type = "token_replace" if i2v else None
        return cos, sin

    def run_steps(m, cos, sin):
        outs = []
        for cnt, tval in steps:
            m.cnt = cnt
            outs.append(m(x, torch.tensor([tval], device=dev), text, mask, text2, cos, sin, gd, return_dict=False))
        return outs

    # ---- single rank, with the multi-GPU top_k rule (models_mul...:249-251): N * int((1 - r) * (S_loc // 128))
    single = copy.deepcopy(base)
    cos, sin = configure(single)
    orig = dit._select_top_k
    dit._select_top_k = lambda r, nblk: N * int((1 - r) * ((nblk * 128 // N) // 128))
    try:
        want = run_steps(single, cos, sin)
    finally:
        dit._select_top_k = orig
    torch.cuda.synchronize()
    assert all(torch.isfinite(w.float()).all() for w in want)
    assert not torch.equal(want[0], want[2])

    # ---- N simulated ranks
    world = SimWorld(N)
    results, errors = [None] * N, []
    models = [copy.deepcopy(base) for _ in range(N)]

    def run(rank):
        try:
            torch.cuda.set_device(dev)
            m = models[rank]
            ulysses.set_thread_sp_group(SimGroup(world, rank))
            ex = SimExchange(world, rank)      # ONE exchange per rank: its call counter orders the collectives
            for blk in list(m.double_blocks) + list(m.single_blocks):
                sp = ulysses.UlyssesAttenCarve(exchange=ex, pipeline=pipe)
                blk.hybrid_seq_parallel_attn = sp if fused else _ReferenceSignatureOnly(sp)
            c, s = configure(m)
            results[rank] = run_steps(m, c, s)
        except Exception as e:                                 # noqa: BLE001 - surfaced below
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            world.barrier.abort()
        finally:
            ulysses.set_thread_sp_group(None)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    torch.cuda.synchronize()

    rec = {"N": N, "latent": list(latent), "S_loc": S_img // N, "i2v": i2v, "fused_prologue": fused, "steps": []}
    for si, (cnt, _) in enumerate(steps):
        # the gathered output is assembled from all ranks' shards, so every rank must hold the same tensor
        for r in range(1, N):
            assert torch.equal(results[r][si], results[0][si]), f"step {cnt}: rank {r} differs from rank 0"
        got, ref = results[0][si].float(), want[si].float()
        exact = bool(torch.equal(results[0][si], want[si]))
        err = (got - ref).abs()
        bound = 2 * torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-3))) - 7) + 0.02
        frac_out = float((err > bound).float().mean().item())
        rec["steps"].append({"cnt": cnt, "bit_exact": exact, "max_abs": float(err.max().item()),
                             "mean_abs": float(err.mean().item()), "frac_beyond_2ulp": frac_out})
        assert exact or (frac_out <= 5e-3 and err.mean().item() <= 3e-3), rec["steps"][-1]
    # the residual cache of a rank is its LOCAL shard (jenga_hyvideo_multigpu.py:296-305)
    for r in range(N):
        assert models[r].previous_residual.shape[1] == S_img // N
    if pipe:      # per-head selection and attention: the pipelined call must reproduce the single-rank forward bit for bit
        assert all(st["bit_exact"] for st in rec["steps"]), rec["steps"]
    _record(f"N{N}_{'i2v' if i2v else 't2v'}_{'fused' if fused else 'unfused'}{'_pipelined' if pipe else ''}", rec)


def test_first_frame_mask_on_a_chunked_order_matches_the_unchunked_mask(dev):
    """dit.py builds first_frame_mask = (order < th * tw) from the rank's CHUNK of hilbert_order: the concatenation of
    the chunks' masks must be the single-rank mask (= mask[:th*tw] = 1 gathered into curve order,
    jenga_hyi2v.py:124-130)."""
    from jenga_amd import gilbert as G
    tt, th, tw = 3, 12, 32
    _, h2l = G.gilbert_mapping(tt, th, tw, as_tensor=True, device=dev)
    full = (h2l < th * tw)
    lin = torch.zeros(tt * th * tw, dtype=torch.bool, device=dev)
    lin[: th * tw] = True
    assert torch.equal(full, lin[h2l])
    for N in (2, 4, 8):
        parts = [(c < th * tw) for c in torch.chunk(h2l, N, dim=0)]
        assert torch.equal(torch.cat(parts), full)
