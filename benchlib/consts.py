"""Part of bench.py (repo root): constants.  Split out of bench.py in round 6;
bench.py re-exports these names."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md
FLOPS_PER_PAIR = 4 * 128 ** 3  # one 128x128 query block against one 128-key block, head_dim 128: QK^T + PV
PMC_FILE = "r05_pmc_bsattn_lp_rates.json"   # the counter passes `roofline.traffic` falls back to (profiles/): per drop rate
ATTN_ALGORITHMIC_BYTES = 4 * 115456 * 24 * 128 * 2 + 24 * 900 * 902 * 4      # Q + K + V + O of one launch + the kept lists (config 2)


def attn_kernel_name():
    """Name of the attention kernel the current default flags launch (as rocprofv3 prints it)."""
    from jenga_amd import _capi
    fl = _capi.ATTN_DEFAULT_FLAGS
    return ("jenga::bsattn_lq_kernel<bf16>" if fl & _capi.ATTN_PAIR else
            "jenga::bsattn_lp_kernel<bf16>" if fl & _capi.ATTN_LP else "jenga::bsattn_fwd_kernel<bf16>")
