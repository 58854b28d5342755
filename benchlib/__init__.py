"""Helpers of bench.py (the driver's benchmark at the repo root): constants, launcher, power sampling, secondary rooflines, the CPU
baseline, the in-run counter passes and the Wan2.1 workload.  bench.py keeps the argument parser, the presets and main()."""
