"""The cost launches of the shipped circuits at scale, on their own (the shipped_circuits_cost_at_scale block of
bench.py): time_shipped_cost.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda", 0)
    for row in bench.bench_shipped_cost(torch, dev, steps):
        print(json.dumps({k: row[k] for k in ("config", "cores", "row_bytes", "partitions", "kernel_ms")} |
                         {"frac": round(row["roofline"]["frac"], 4), "cost0": row["cost_of_the_block_assignment_run"],
                          "ref": row["reference_gtrhqa_cost"]}), flush=True)


if __name__ == "__main__":
    main()
