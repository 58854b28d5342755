"""Part of bench.py (repo root): the self-launcher for --gpus N and the stdout guard around RCCL's banner.  Split out of bench.py in round 6;
bench.py re-exports these names."""
import os
import sys

from .consts import ROOT


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N` re-executes itself as when no launcher set WORLD_SIZE."""
    if port is None:
        port = _free_port()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


class _StdoutToStderr:
    """File descriptor 1 -> stderr until restore(); libc's and Python's buffers are flushed on both edges."""

    def __init__(self):
        sys.stdout.flush()
        _flush_c_stdio()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def restore(self):
        if self.saved is None:
            return
        sys.stdout.flush()
        _flush_c_stdio()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        self.saved = None


def _flush_c_stdio():
    """fflush(NULL): whatever native libraries left in libc's stdio buffers goes out now (see the end of main)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:               # noqa: BLE001
        pass
