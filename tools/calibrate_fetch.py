#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration for THIS code's access pattern (4-byte-per-lane coalesced
loads and stores), as MI355X_MICROARCH.md asks before trusting an absolute HBM byte count:
k_disp2depth reads and writes a known number of bytes, once each, far beyond the 256 MiB L3.
Run under rocprofv3 --pmc FETCH_SIZE (and again with WRITE_SIZE); tools/pmc_summary.py prints the
per-launch counters to compare with the known byte count printed here."""
import sys

import numpy as np

sys.path.insert(0, ".")
import svo_loader

pkg = svo_loader.load()
n = 900000                 # 3.6 MB in + 3.6 MB out per call (fits the host API scratch), 64 calls
svo = pkg.Svo(640, 240)
disp = np.random.default_rng(0).integers(0, 49, n).astype(np.float32)
for _ in range(64):
    svo.disp2depth(disp, 386.1448)
print("k_disp2depth: %d launches, %d bytes read and %d bytes written per launch" % (64, 4 * n, 4 * n))
