"""CPU: host logic of the Wan driver (drop-rate warm-up, TeaCache decision) against hand-derived sequences that follow
jenga_wan.py:190-206 and :595-626 line by line."""
import numpy as np
import torch

from jenga_amd.wan_driver import TEACACHE_COEFFS, TeaCache, sa_drop_rate_for_step


def test_drop_rate_schedule():
    rates = [0.7, 0.8]
    n = 50
    got = [sa_drop_rate_for_step(i, n, rates) for i in range(n)]
    assert got[0] == 0.0                                   # warm-up starts dense
    assert abs(got[1] - (1 / 49 * 10) * 0.7) < 1e-12
    assert got[5] == 0.7 and got[25] == 0.7                # 5/49*10 > 1 -> capped at the rate
    assert got[26] == 0.8 and got[49] == 0.8
    assert sa_drop_rate_for_step(30, n, [0.6]) == 0.6


def _ref_decisions(embs, num_steps, thresh, coeffs, ret_steps, cutoff):
    """Independent restatement of the even/odd accumulators."""
    acc = {0: 0.0, 1: 0.0}
    prev = {}
    out = []
    for cnt, e in enumerate(embs):
        p = cnt % 2
        if cnt < ret_steps or cnt >= cutoff:
            calc = True
            acc[p] = 0.0
        else:
            rel = float((e - prev[p]).abs().mean() / prev[p].abs().mean())
            acc[p] += float(np.poly1d(coeffs)(rel))
            calc = not (acc[p] < thresh)
            if calc:
                acc[p] = 0.0
        prev[p] = e
        out.append(calc)
    return out


def test_teacache_decisions_follow_the_reference_rule():
    torch.manual_seed(0)
    steps = 12
    base = torch.randn(1, 64)
    embs = [base * (1 + 0.02 * (i // 2)) + 0.001 * torch.randn(1, 64) for i in range(2 * steps)]
    tea = TeaCache(steps, thresh=0.08, task="t2v-1.3B", use_ret_steps=False)
    got = []
    for e in embs:
        calc, parity = tea.decide(e, e.unsqueeze(1).repeat(1, 6, 1))
        assert parity == tea.cnt % 2
        got.append(calc)
        tea.advance()
    ref = _ref_decisions(embs, steps, 0.08, TEACACHE_COEFFS[("t2v-1.3B", False)], 2, 2 * steps - 2)
    assert got == ref
    assert got[0] and got[1] and got[-1] and got[-2]          # first step and last step are always computed
    assert not all(got)                                       # something was actually skipped
    assert tea.cnt == 0                                       # wrapped around after num_steps*2 calls


def test_teacache_stage_start_forces_compute():
    tea = TeaCache(10, thresh=10.0)
    e = torch.ones(1, 8)
    for _ in range(4):
        tea.decide(e, e.unsqueeze(1))
        tea.advance()
    calc, _ = tea.decide(e, e.unsqueeze(1))
    assert not calc                                           # huge threshold: skipped
    tea.advance()
    tea.stage_start = True
    calc, _ = tea.decide(e, e.unsqueeze(1))
    assert calc


def test_wan_flow_schedule_and_turbo_switch_match_the_reference_scheduler(golden_dir):
    """tests/golden/wan_sched_cases.npz: FlowUniPCMultistepScheduler's own sigmas / timesteps and the Turbo stage switch
    composed as jenga_wan.py:217-243 orders its step_to_zero / add_noise / set_timesteps calls."""
    import os
    from jenga_amd.wan_driver import WAN_TURBO_DISABLE_CORRECTOR, WanFlowSchedule, wan_switch_stage
    g = np.load(os.path.join(golden_dir, "wan_sched_cases.npz"))
    for shift in (3.0, 5.0):
        s = WanFlowSchedule(50, shift)
        assert np.array_equal(s.sigmas.numpy(), g[f"sigmas_{int(shift)}"])
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{int(shift)}"])
    s = WanFlowSchedule(50, 3.0)
    lat, npred, noise = (torch.from_numpy(g[k]) for k in ("lat", "npred", "noise"))
    out = wan_switch_stage(s, npred, 25, lat, (3, 8, 12), noise, 50)
    assert np.array_equal(out.numpy(), g["switched"])
    assert np.array_equal(s.sigmas.numpy(), g["sigmas_after"]) and np.array_equal(s.timesteps.numpy(), g["timesteps_after"])
    assert s.shift == 5.0 and s.disable_corrector == WAN_TURBO_DISABLE_CORRECTOR
    # the corrector gate right after the switch (fm_solvers_unipc.py:688-692, 723-725)
    assert not s.use_corrector(26) and s.use_corrector(23) and not s.use_corrector(0)
    assert s.order_after_gate(2) == 1 and s.disable_corrector == [] and s.order_after_gate(2) == 2
    assert s.use_corrector(26)
