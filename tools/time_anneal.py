"""Times the annealing leg of bench.py on its own: time_anneal.py [sweeps] [replicas]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import bench_qubo  # noqa: E402

if __name__ == "__main__":
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    out = bench_qubo(torch, torch.device("cuda", 0), 0, sweeps, R)
    print({k: out[k] for k in ("spin_flip_attempts_per_s", "anneal_s", "best_energy", "energy_ms")})
