#!/usr/bin/env python3
"""Compact view of a bench_attn.py --also-flags record or a bench.py line (GPU session logs)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if "attn_ms" in d:
    print("lists: shared %.3f kept %.1f pairs %d" % (d.get("adjacent_shared_frac", -1), d["kept_mean"], d["pairs"]))
    print("  flags %-4s %7.2f ms %6.0f TF %.4f   (second pass %.2f ms)" % (d["flags"], d["attn_ms"], d["attn_TFLOPs"],
                                                                         d["attn_frac_of_2.5PF"], d.get("attn_ms_second_pass", 0)))
    for k, v in d.get("also", {}).items():
        print("  flags %-4s %7.2f ms %6.0f TF %.4f   max|d| %.4f mean|d| %.2e finite %s" % (
            k, v["attn_ms"], v["attn_TFLOPs"], v["frac"], v["max_abs_diff_vs_first"], v["mean_abs_diff_vs_first"], v["finite"]))
else:
    r = d["roofline"]
    print("value %.2f s/video  ms/step %.1f  kernel %s  frac %.4f  avg launch %.3f ms  shared %.3f" % (
        d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["avg_launch_ms"], r.get("adjacent_shared_frac", -1)))
    o = d.get("extra", {}).get("attn_other_kernel")
    if o:
        print("other kernel: est %.2f s/video  frac %.4f  avg launch %.3f ms  (%s)" % (
            o["s_per_video_estimate"], o["attention_frac_of_peak"], o["attention_avg_launch_ms"], o["what"][:40]))
