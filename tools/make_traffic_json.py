#!/usr/bin/env python3
"""HBM bytes per launch of the headline cost kernels from the two PMC passes tools/profile_bench.sh made
(gpurun_out/<tag>_{uniform,integer,integer_table}_hbm_traffic.csv): one JSON entry per kernel name, as bench.py reads them.
FETCH_SIZE (KB) is doubled: on gfx950 it reports half the bytes of a 16-byte-per-lane streaming read."""
import json
import sys

tag = sys.argv[1]
KERNEL = {"uniform": "k_cost_v3_pairs<4, false>", "integer": "k_cost_v3_grid_row<false, 4>",
          "integer_table": "k_cost_v3_row4<1, 4>"}
out = {}
for w, kernel in KERNEL.items():
    vals = {}
    path = f"gpurun_out/{tag}_{w}_hbm_traffic.csv"
    for line in open(path):
        name, avg, n = line.strip().split(",")
        vals[name] = float(avg)
    fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    out[kernel] = {"Q": 1024, "T": 1000, "B": 4096, "source": "gtrhqa", "weights": w.replace("_table", ""),
                   "fetch_size_kb_reported": fetch_kb, "write_size_kb": write_kb, "fetch_correction": 2.0,
                   "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024)),
                   "profile": f"profiles/{tag}_{w}_hbm_traffic.csv"}
print(json.dumps(out, indent=1))
