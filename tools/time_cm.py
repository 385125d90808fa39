"""Times the cost-matrix kernel on the benchmark circuit for several batch sizes: time_cm.py [B ...]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import bench_cost_matrix  # noqa: E402
from quantum_simulator_amd import layered_gate_arrays  # noqa: E402
from quantum_simulator_amd.engine import DeviceCircuit, apsp  # noqa: E402
from quantum_simulator_amd.multicore import graph_tables, multicore  # noqa: E402
from quantum_simulator_amd.timeslices import CircuitIR  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    q0, q1, _ = layered_gate_arrays(1024, 1000, 0.5, seed=2025)
    ir = CircuitIR.from_gate_tables(1024, q0, q1)
    dc = DeviceCircuit(ir, dev)
    random.seed(1234)
    arch = multicore(16, 16, ncm_qubits=4, ncp_qubits=8, edge_weight_range=(2.0, 2.0), network_type="mesh")
    nodes, eu, ev, w = graph_tables(arch.network)
    D = apsp(len(nodes), eu, ev, w)
    for B in [int(x) for x in sys.argv[1:]] or [256, 1024, 2048, 8192]:
        r = bench_cost_matrix(torch, dev, dc, ir, D, B, 20)
        print(f"B={B}: {r['kernel_ms']:.4f} ms  {r['entries_per_s']:.3e} entries/s  {r['roofline']['achieved']:.0f} GB/s written", flush=True)


if __name__ == "__main__":
    main()
