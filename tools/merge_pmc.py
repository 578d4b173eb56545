#!/usr/bin/env python3
"""profiles/pmc_latest.json from the per-workload PMC summaries of tools/collect_profiles.sh:
   python tools/merge_pmc.py <track_pmc.json> <frontend_pmc.json> [<legs_pmc.json> ...]
Front-end kernels are annotated with the pairs one dispatch covered when the counters were taken (the frontend workload runs
128 pairs per step as two slices of 64), ELAS kernels with their chunk of 32 pairs, so that bench.py can scale
FETCH_SIZE / WRITE_SIZE to the batch it runs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRONT = ("k_pyr_fused", "k_pyr_level", "k_fast", "k_select", "k_describe", "k_stereo_match", "k_stereo_median")
out = {}
track = json.load(open(sys.argv[1]))
front = json.load(open(sys.argv[2]))
legs = {}
for f in sys.argv[3:]:
    legs.update(json.load(open(f)))
for k, v in legs.items():                       # traffic counters only (FETCH_SIZE / WRITE_SIZE passes)
    if k.startswith(("k_elas", "k_cc_")):
        v["_pairs_per_dispatch_traffic"] = 32
    out[k] = v
for k, v in track.items():                      # the tail kernels (and the front end at 32 pairs per dispatch)
    if k.startswith("k_probe_"):                # (the stream probe of the first tracker call)
        continue
    if k in FRONT:
        v["_pairs_per_dispatch_traffic"] = 32
    out[k] = v
for k, v in front.items():
    if k in FRONT:
        v["_pairs_per_dispatch_traffic"] = 64
        out[k] = v
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1, sort_keys=True)
print(sorted(out))
