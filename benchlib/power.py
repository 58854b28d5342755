"""Part of bench.py (repo root): host CPU description and the board power / clock sampler.  Split out of bench.py in round 6;
bench.py re-exports these names."""
import os
import sys
import time


def _host_cpu():
    """(model name, physical cores, logical cpus) of the box this runs on."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    physical = logical
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        pass
    return model, physical, logical


class PowerSampler:
    """Board power / shader clock from the amdgpu hwmon files, sampled by a host thread while the timed region runs
    (the kernels sit on the 1400 W board cap, so every throughput figure in this file is a figure AT a power / clock
    state: DESIGN.md 3 K1 point 4).  None of it touches the GPU queues."""

    def __init__(self, period=0.25):
        import glob
        self.period, self.samples, self.nodes = period, [], []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            f = {k: f"{d}/{k}" for k in ("power1_average", "power1_input", "power1_cap", "freq1_input")}
            if self._read(f["power1_average"]) is not None or self._read(f["power1_input"]) is not None:
                self.nodes.append(f)
        self._stop = None
        self._thr = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return int(fh.read().strip())
        except Exception:   # noqa: BLE001
            return None

    def _loop(self):
        while not self._stop.is_set():
            best = None
            for f in self.nodes:      # the busiest card (a box may expose more hwmon nodes than HIP devices)
                pw = self._read(f["power1_average"])
                if pw is None:
                    pw = self._read(f["power1_input"])
                if pw is not None and (best is None or pw > best[0]):
                    best = (pw, self._read(f["freq1_input"]) or 0)
            if best:
                self.samples.append((best[0] / 1e6, best[1] / 1e6))
            self._stop.wait(self.period)

    def start(self):
        import threading
        if self.nodes:
            self._stop = threading.Event()
            self._thr = threading.Thread(target=self._loop, daemon=True)
            self._thr.start()
        return self

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=2)
        if not self.samples:
            return {"available": False, "note": "no amdgpu hwmon power node readable on this box"}
        pw = sorted(s_[0] for s_ in self.samples)
        fq = sorted(s_[1] for s_ in self.samples if s_[1] > 0)
        out = {"available": True, "samples": len(pw), "period_s": self.period,
               "power_W": {"mean": round(sum(pw) / len(pw), 1), "median": round(pw[len(pw) // 2], 1),
                           "max": round(pw[-1], 1)},
               "power_cap_W": (self._read(self.nodes[0]["power1_cap"]) or 0) / 1e6,
               "source": "amdgpu hwmon power1_average|power1_input / freq1_input over the timed region"}
        if fq:
            out["sclk_MHz"] = {"mean": round(sum(fq) / len(fq)), "median": round(fq[len(fq) // 2]), "min": round(fq[0]),
                               "max": round(fq[-1])}
        return out
