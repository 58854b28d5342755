"""ProRes (progressive resolution) denoising-loop glue, counterpart of
   hyvideo/diffusion/schedulers/scheduling_flow_match_discrete.py   set_timesteps :124-156, sd3_time_shift :185-186,
                                                                     step :188-254, predict_x0_from_xt :259-282,
                                                                     add_noise_to_step :284-299
   hyvideo/diffusion/pipelines/pipeline_hunyuan_video_prores.py     stage shapes :418-423, text_amp :577,
                                                                     stage switch :697-767

O(latent) scalar math on tensors of a few MB: plain torch on whatever device the latents live on (SURVEY.md marks the
scheduler file itself out of scope; the stage switch is row f-1 because it decides which curve / RoPE table / drop
rate / text_amp the hot path sees)."""
import math

import torch


def sd3_time_shift(t, shift):
    return (shift * t) / (1 + (shift - 1) * t)


class FlowMatchSchedule:
    """Euler flow-matching schedule with the Jenga additions (reverse=True as `--flow-reverse` sets it)."""

    def __init__(self, num_inference_steps, shift=7.0, reverse=True, num_train_timesteps=1000):
        self.num_train_timesteps = num_train_timesteps
        self.reverse = reverse
        self.set_timesteps(num_inference_steps, shift)

    def set_timesteps(self, num_inference_steps, shift):
        self.shift = shift
        sigmas = sd3_time_shift(torch.linspace(1, 0, num_inference_steps + 1), shift)
        if not self.reverse:
            sigmas = 1 - sigmas
        self.sigmas = sigmas
        self.timesteps = (sigmas[:-1] * self.num_train_timesteps).to(torch.float32)

    def step(self, noise_pred, i, latents):
        dt = self.sigmas[i + 1] - self.sigmas[i]
        return latents.to(torch.float32) + noise_pred.to(torch.float32) * dt

    def predict_x0_from_xt(self, noise_pred, i, latents):
        d_sigma = self.sigmas[-1] - self.sigmas[i]
        return latents.to(torch.float32) + noise_pred.to(torch.float32) * d_sigma

    def add_noise_to_step(self, latents, noise, i):
        s = self.sigmas[i]
        return latents.to(torch.float32) * (1.0 - s) + noise.to(torch.float32) * s


def stage_plan(latent_thw, num_steps, res_rate_list, step_rate_list):
    """-> (latent shapes per stage [(T, H', W')] in latent pixels, switch steps).  Mirrors the pipeline's
    int(height*rate) // 16 * 2 arithmetic on PIXEL sizes (:418-423, 571-575): latent H = pixel H / 8."""
    T, H, W = latent_thw
    ph, pw = H * 8, W * 8
    shapes = [(T, int(ph * r) // 16 * 2, int(pw * r) // 16 * 2) for r in res_rate_list]
    split = [int(num_steps * s) for s in step_rate_list]
    return shapes, split


def stage_text_amp(stage_shape, final_shape, scale_txt_amp=1.0):
    """text_amp of a reduced-resolution stage: -log2(sqrt(tokens/tokens_final)) * scale (:577, 594)."""
    tok = (stage_shape[1] // 2) * (stage_shape[2] // 2)
    tok_f = (final_shape[1] // 2) * (final_shape[2] // 2)
    return -1 * math.log(math.sqrt(tok / tok_f), 2) * scale_txt_amp


def stage_text_amps(shapes, scale_txt_amp=1.0):
    """text_amp per stage: only stage 0 carries the amplifier (:577, 594); the pipeline sets text_amp = 0.0 after ANY
    stage switch (:755), so every later stage runs with 0.0 whatever its resolution."""
    return [stage_text_amp(shapes[0], shapes[-1], scale_txt_amp)] + [0.0] * (len(shapes) - 1)


def switch_stage(sched, noise_pred, i, latents, new_shape, new_shift, noise):
    """The re-noising hop at a stage boundary (:724-739): re-shift the schedule, predict x0 from x_t, upsample
    trilinearly to the next stage's latent size, add the fresh noise at sigma_{i+1}.  Returns fp32 latents."""
    sched.set_timesteps(len(sched.sigmas) - 1, new_shift)
    x0 = sched.predict_x0_from_xt(noise_pred, i, latents)
    x0 = torch.nn.functional.interpolate(x0, size=list(new_shape), mode="trilinear")
    return sched.add_noise_to_step(x0, noise, i + 1)
