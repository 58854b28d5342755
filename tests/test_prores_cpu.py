"""CPU: ProRes loop glue (row f-1) against goldens from the reference's FlowMatchDiscreteScheduler and the pipeline's
stage arithmetic (pipeline_hunyuan_video_prores.py:418-423, 577, 697-767)."""
import os

import numpy as np
import torch

from jenga_amd import prores


def test_schedule_matches_reference_scheduler(golden_dir):
    g = np.load(os.path.join(golden_dir, "scheduler_cases.npz"))
    lat, npred, noise = (torch.from_numpy(g[k]).to(torch.bfloat16) for k in ("lat", "npred", "noise"))
    for shift in (7.0, 9.0):
        s = prores.FlowMatchSchedule(50, shift=shift)
        assert np.array_equal(s.sigmas.numpy(), g[f"sigmas_{int(shift)}"])
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{int(shift)}"])
        assert np.array_equal(s.predict_x0_from_xt(npred, 25, lat).numpy(), g[f"x0_{int(shift)}"])
        assert np.array_equal(s.add_noise_to_step(lat, noise, 26).numpy(), g[f"renoise_{int(shift)}"])
        assert np.array_equal(s.step(npred, 25, lat).numpy(), g[f"step_{int(shift)}"])


def test_stage_plan_and_text_amp():
    # Turbo: --res-rate-list 0.75 1.0 --step-rate-list 0.5 1.0 on 720x1280x125f (latent 32x90x160)
    shapes, split = prores.stage_plan((32, 90, 160), 50, [0.75, 1.0], [0.5, 1.0])
    assert shapes == [(32, 66, 120), (32, 90, 160)] and split == [25, 50]
    assert (shapes[0][1] // 2) * (shapes[0][2] // 2) * 32 == 63360          # 495 blocks
    amp = prores.stage_text_amp(shapes[0], shapes[-1])
    assert abs(amp - 0.431) < 1e-3                                            # -log2(sqrt(1980/3600))
    shapes3, split3 = prores.stage_plan((32, 90, 160), 50, [0.5, 0.75, 1.0], [0.3, 0.5, 1.0])
    assert shapes3[0] == (32, 44, 80) and split3 == [15, 25, 50]
    assert 32 * 22 * 40 == 28160
    # only stage 0 carries the amplifier: the pipeline zeroes text_amp after ANY switch (pipeline...prores.py:755),
    # so the 0.75-resolution middle stage of the 3-stage presets runs with 0.0
    amps3 = prores.stage_text_amps(shapes3)
    assert amps3[1:] == [0.0, 0.0] and abs(amps3[0] - 1.0162107388461887) < 1e-12     # -log2(sqrt(880/3600))
    assert [abs(v) for v in prores.stage_text_amps(prores.stage_plan((32, 90, 160), 50, [1.0, 1.0], [0.5, 1.0])[0])] == [0.0, 0.0]


def test_switch_stage_composes_the_three_steps():
    torch.manual_seed(0)
    s = prores.FlowMatchSchedule(50, shift=7.0)
    lat = torch.randn(1, 4, 3, 4, 6)
    npred = torch.randn(1, 4, 3, 4, 6)
    noise = torch.randn(1, 4, 3, 6, 8)
    out = prores.switch_stage(s, npred, 25, lat, (3, 6, 8), 9.0, noise)
    ref = prores.FlowMatchSchedule(50, shift=9.0)
    x0 = torch.nn.functional.interpolate(ref.predict_x0_from_xt(npred, 25, lat), size=[3, 6, 8], mode="trilinear")
    assert torch.equal(out, ref.add_noise_to_step(x0, noise, 26)) and s.shift == 9.0


def test_switch_stage_matches_the_reference_composition(golden_dir):
    """tests/golden/stage_switch_case.npz: the reference scheduler's own predict_x0_from_xt -> trilinear interpolate ->
    add_noise_to_step after the re-shift, in the order pipeline_hunyuan_video_prores.py:724-739 calls them, plus the
    per-stage quantities the switch swaps (latent shapes :423-424/:705, split steps :422, text_amp :577/:594/:755)."""
    g = np.load(os.path.join(golden_dir, "stage_switch_case.npz"))
    shifts = g["shifts"].tolist()
    lat, npred, noise = (torch.from_numpy(g[k]).to(torch.bfloat16) for k in ("lat", "npred", "noise"))
    lat_shapes = [tuple(x) for x in g["lat_shapes"].tolist()]
    shapes, split = prores.stage_plan(lat_shapes[-1], 50, [0.75, 1.0], [0.5, 1.0])
    assert shapes == lat_shapes and split == g["split"].tolist()
    assert abs(prores.stage_text_amp(shapes[0], shapes[-1]) - float(g["text_amp_stage0"])) < 1e-12
    s = prores.FlowMatchSchedule(50, shift=shifts[0])
    out = prores.switch_stage(s, npred, split[0], lat, shapes[1], shifts[1], noise)
    assert np.array_equal(s.sigmas.numpy(), g["sigmas_after"])
    assert np.array_equal(out.numpy(), g["switched"])
