#!/usr/bin/env python3
"""Board power and shader clock while a command runs (GPU box):
    python tools/power_sample.py --out gpurun_out/x.json -- python tools/bench_attn.py --iters 200 --attn-only ...
Polls the amdgpu hwmon files (power1_average / power1_input in uW, freq1_input in Hz, power1_cap) every 50 ms and, if
they are absent, `rocm-smi --showpower --showclocks --json` every 0.5 s.  Prints / writes the time series summary:
idle level (first second), the busy plateau (top half of the samples) and the cap."""
import argparse
import glob
import json
import subprocess
import sys
import time


def _read(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:  # noqa: BLE001
        return None


def hwmon():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        files = {k: f"{d}/{k}" for k in ("power1_average", "power1_input", "power1_cap", "freq1_input")}
        if _read(files["power1_average"]) is not None or _read(files["power1_input"]) is not None:
            out.append(files)
    return out


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True,
                           text=True, timeout=5)
        return json.loads(r.stdout)
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def main():
    if "--" not in sys.argv:
        sys.exit(__doc__)
    k = sys.argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--period", type=float, default=0.05)
    a = ap.parse_args(sys.argv[1:k])
    cmd = sys.argv[k + 1:]
    hw = hwmon()
    res = {"cmd": " ".join(cmd), "hwmon": bool(hw), "smi_before": smi()}
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    series = []
    smi_series = []
    smi_mid = None
    while p.poll() is None:
        t = time.time() - t0
        if hw:
            # the busiest card (a box may expose more hwmon nodes than HIP devices)
            best = None
            for f in hw:
                pw = _read(f["power1_average"])
                if pw is None:
                    pw = _read(f["power1_input"])
                if pw is not None and (best is None or pw > best[0]):
                    best = (pw, (_read(f["freq1_input"]) or 0))
            series.append((round(t, 3), best[0] / 1e6 if best else None, (best[1] if best else 0) / 1e6))
            if len(series) % 40 == 0:
                smi_series.append((round(t, 1), smi()))
            time.sleep(a.period)
        else:
            s = smi()
            series.append((round(t, 3), s))
            time.sleep(0.5)
        if smi_mid is None and t > 20 and hw:
            smi_mid = smi()
    so, se = p.communicate()
    res["rc"] = p.returncode
    res["stdout_tail"] = so[-1500:]
    res["stderr_tail"] = se[-500:]
    res["smi_mid_run"] = smi_mid
    res["smi_series"] = smi_series
    res["hwmon_nodes"] = len(hw)
    if hw:
        res["power_cap_W"] = (_read(hw[0]["power1_cap"]) or 0) / 1e6
        pw = [s[1] for s in series if s[1] is not None]
        fq = [s[2] for s in series if s[2]]
        if pw:
            srt = sorted(pw)
            res["power_W"] = {"n": len(pw), "min": srt[0], "median": srt[len(srt) // 2], "p90": srt[int(len(srt) * 0.9)],
                              "max": srt[-1]}
        if fq:
            srt = sorted(fq)
            res["sclk_MHz"] = {"min": srt[0], "median": srt[len(srt) // 2], "p10": srt[int(len(srt) * 0.1)], "max": srt[-1]}
        res["series_every_10th"] = series[::10]
    else:
        res["series"] = series
    txt = json.dumps(res, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
    print(txt[-3000:])


if __name__ == "__main__":
    main()
