"""hipBLASLt solution selection for the DiT's dense GEMMs (QKV / proj / MLP: torch.nn.functional.linear ->
hipBLASLt; outside the AttenCarve hot path).

hipBLASLt's default heuristic is not the fastest pick for every shape of this model (round 3, MI355X, the per-rank
shapes of an 8-rank Ulysses job, M = 14400: qkv 0.704 -> 0.583 ms, fc2 0.936 -> 0.845 ms, linear2 1.289 -> 1.084 ms).
PyTorch's TunableOp can time every solution the library offers for a shape and record the winner; this module only
(1) records such a file (tune=True) and (2) replays a committed one with tuning OFF (no timing at run time, shapes
that are not in the file use the default pick).  The files are plain TunableOp CSVs and carry TunableOp's own
validators (torch / ROCm / hipBLASLt versions, gfx arch): on another stack they are ignored.
"""
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FILE = os.path.join(HERE, "tuned", "hipblaslt_gfx950.csv")


def enable(path=None, tune=False):
    """Replay (tune=False) or record (tune=True) a TunableOp selection file.  Returns the path used, or None if there
    is nothing to replay."""
    path = path or DEFAULT_FILE
    if not tune and not os.path.exists(path):
        return None
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(bool(tune))
    t.set_filename(path, insert_device_ordinal=False)
    if not tune:
        try:
            ok = t.read_file(path)
        except Exception:          # noqa: BLE001 - a selection file is an optimisation, never a reason to fail
            ok = False
        if not ok:                 # validators do not match this stack: fall back to the library's heuristic
            t.enable(False)
            return None
    return path


def disable():
    torch.cuda.tunable.enable(False)
