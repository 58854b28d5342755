"""Wan2.1 Jenga driver pieces, counterpart of jenga_wan.py:
   drop-rate schedule with warm-up        t2v_generate :190-206
   TeaCache skip decision (even = cond, odd = uncond CFG stream)   teacache_forward :595-626
   forward skeleton: pad -> Hilbert gather -> (blocks | cached residual) -> scatter    teacache_forward :545-659

   Turbo stage switch + the UniPC hooks it drives               t2v_generate :217-243, fm_solvers_unipc.py:107-117,
                                                                 182-211 (schedule), 688-692 / 723-725 (corrector gate),
                                                                 805-870 (step_to_zero), 761-803 (add_noise)

The skeleton takes the blocks as callables (jenga_amd.modules.wan.WanSelfAttention-based blocks in the tests) because
the full Wan model (T5 / CLIP / VAE / cross-attention zoo) is out of the hot-path scope; gather/scatter run the HIP
row-gather kernel, the decision logic is scalar host code on the [B, 6, dim] modulation embedding.
"""
import numpy as np
import torch

from . import _capi

# (rescale polynomial, ret_steps, cutoff rule) for t2v, jenga_wan.py:1085-1098
TEACACHE_COEFFS = {
    ("t2v-1.3B", True): [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02],
    ("t2v-14B", True): [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01],
    ("t2v-1.3B", False): [2.39676752e+03, -1.31110545e+03, 2.01331979e+02, -8.29855975e+00, 1.37887774e-01],
    ("t2v-14B", False): [-5784.54975374, 5449.50911966, -1811.16591783, 256.27178429, -13.02252404],
}


def sa_drop_rate_for_step(idx, n_steps, rates):
    """jenga_wan.py:190-206: rate[0] up to step 25, rate[1] after; linear warm-up min(rate, idx/(n-1)*10*rate)."""
    rate = rates[0] if (idx <= 25 or len(rates) == 1) else rates[1]
    step_normed = idx / (n_steps - 1) * 10
    return min(rate, step_normed * rate)


class TeaCache:
    """State machine of teacache_forward's skip decision.  One instance per model; `cnt` counts forward calls
    (2 per denoising step: even = conditional, odd = unconditional), each parity has its own accumulator."""

    def __init__(self, num_steps, thresh, task="t2v-1.3B", use_ret_steps=False, enable=True):
        self.enable = enable
        self.thresh = thresh
        self.use_ref_steps = use_ret_steps
        self.coefficients = TEACACHE_COEFFS[(task, use_ret_steps)]
        self.num_steps = num_steps * 2
        if use_ret_steps:
            self.ret_steps, self.cutoff_steps = 5 * 2, num_steps * 2
        else:
            self.ret_steps, self.cutoff_steps = 1 * 2, num_steps * 2 - 2
        self.cnt = 0
        self.stage_start = False
        self.acc = [0.0, 0.0]
        self.prev = [None, None]
        self.residual = [None, None]

    def decide(self, e, e0):
        """e [B, dim] / e0 [B, 6, dim] time embeddings of this call -> (should_calc, parity)."""
        parity = self.cnt % 2
        if not self.enable:
            return True, parity
        inp = e0 if self.use_ref_steps else e
        if self.cnt < self.ret_steps or self.cnt >= self.cutoff_steps or self.stage_start:
            calc = True
            self.acc[parity] = 0.0
        else:
            rel = ((inp - self.prev[parity]).abs().mean() / self.prev[parity].abs().mean()).cpu().item()
            self.acc[parity] += float(np.poly1d(self.coefficients)(rel))
            if self.acc[parity] < self.thresh:
                calc = False
            else:
                calc = True
                self.acc[parity] = 0.0
        self.prev[parity] = inp.clone()
        return calc, parity

    def advance(self):
        self.cnt += 1
        if self.cnt >= self.num_steps:
            self.cnt = 0


@torch.no_grad()
def teacache_forward(tokens, t_emb, t_emb0, blocks, tea, hilbert_order, linear_to_hilbert, seq_len=None,
                     **block_kwargs):
    """tokens [B, L, C] patch embeddings (L = f*h*w), t_emb [B, dim] / t_emb0 [B, 6, dim] the time embeddings the skip
    decision looks at (block_kwargs may carry its own `e`); pads to seq_len, gathers into curve order, runs the blocks or adds
    the cached residual of this CFG stream, scatters back.  Returns [B, seq_len, C]."""
    B, L, C = tokens.shape
    seq_len = seq_len or L
    if seq_len > L:
        tokens = torch.cat([tokens, tokens.new_zeros(B, seq_len - L, C)], dim=1)
    x = _capi.gather_rows(tokens.contiguous(), hilbert_order)
    calc, parity = tea.decide(t_emb, t_emb0)
    if not calc:
        x = x + tea.residual[parity]
    else:
        ori = x
        for blk in blocks:
            x = blk(x, **block_kwargs)
        if tea.enable:
            tea.residual[parity] = x - ori
    tea.advance()
    return _capi.gather_rows(x.contiguous(), linear_to_hilbert), calc


# ------------------------------------------------------------------------------------------------ Turbo stage switch
class WanFlowSchedule:
    """sigma / timestep schedule of FlowUniPCMultistepScheduler exactly as jenga_wan.py builds it (constructor shift=1,
    then set_timesteps(steps, shift=...), :138-144) plus the two scheduler methods the Turbo stage switch calls.  The
    multistep predictor / corrector updates themselves are the sampler (out of scope); what the hot
    path depends on is WHEN the stage switches and what the corrector gate does right after (below)."""

    def __init__(self, num_steps, shift, num_train_timesteps=1000):
        self.num_train_timesteps = num_train_timesteps
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()     # :107-108
        sig = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)                            # :109-110 (shift 1)
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()                          # :131-132
        self.disable_corrector = []
        self.step_index = None
        self.set_timesteps(num_steps, shift)

    def set_timesteps(self, num_steps, shift):
        sigmas = np.linspace(self.sigma_max, self.sigma_min, num_steps + 1).copy()[:-1]         # :183-185
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)                                    # :192-193
        timesteps = sigmas * self.num_train_timesteps                                           # :205
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0]]).astype(np.float32))        # :206-209 ("zero")
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.int64)                      # :210-211
        self.shift = shift

    def step_to_zero(self, model_output, idx, sample):
        """x0 prediction at the current step (predict_x0, flow_prediction): :838-853."""
        x0 = sample - self.sigmas[idx].to(sample.device) * model_output
        return x0.to(sample.dtype)

    def add_noise(self, original, noise, idx):
        """alpha_t x0 + sigma_t noise at schedule index idx, sigmas cast to the sample dtype first (:768-769, 795-803)."""
        sigma = self.sigmas.to(device=original.device, dtype=original.dtype)[idx]
        return (1 - sigma) * original + sigma * noise

    # ---- the corrector gate (state the stage switch manipulates) ----
    def use_corrector(self, step_index, has_last_sample=True):
        """:688-692"""
        return step_index > 0 and (step_index - 1) not in self.disable_corrector and has_last_sample

    def order_after_gate(self, this_order):
        """:723-725: the FIRST step() after the list was set runs at order 1 and clears the list."""
        if len(self.disable_corrector) > 0:
            self.disable_corrector = []
            return 1
        return this_order


WAN_TURBO_DISABLE_CORRECTOR = list(range(24, 38))      # jenga_wan.py:237


def wan_switch_stage(sched, noise_pred, idx, latents, target_thw, noise, steps):
    """The Turbo resolution hop of jenga_wan.py:217-243 at loop index idx (the reference takes it at idx >= 25):
    x0 = step_to_zero -> trilinear interpolate to the next stage's latent size -> add_noise at timesteps[idx+1] (on the
    OLD shift's sigmas) -> corrector disabled -> schedule re-set with shift + 2.  noise_pred / latents [C,T,H,W];
    returns the new latents [C,T',H',W'] (the caller swaps curve_sels[1], p_remain_rates and sets stage_start, which
    is what makes TeaCache compute the next forward, TeaCache.stage_start)."""
    clean = sched.step_to_zero(noise_pred.unsqueeze(0), idx, latents.unsqueeze(0))
    clean = torch.nn.functional.interpolate(clean, size=list(target_thw), mode="trilinear")
    noisy = sched.add_noise(clean, noise.unsqueeze(0), idx + 1)
    sched.disable_corrector = list(WAN_TURBO_DISABLE_CORRECTOR)
    sched.set_timesteps(steps, sched.shift + 2)
    return noisy.squeeze(0)
