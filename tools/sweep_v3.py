"""Dense cost kernel across circuit widths, core counts and weight kinds (which launcher path each shape takes
and what it reaches): sweep_v3.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quantum_simulator_amd import layered_gate_arrays  # noqa: E402
from quantum_simulator_amd.engine import DeviceCircuit, DistanceTable, Engine  # noqa: E402
from quantum_simulator_amd.multicore import multicore  # noqa: E402
from quantum_simulator_amd.timeslices import CircuitIR  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    eng = Engine(dev)
    print(f"{'Q':>5} {'T':>5} {'mesh':>7} {'weights':>8} {'table':>22} {'ms':>8} {'G evals/s':>10} {'GB/s':>7}")
    # (qubits, slices): the first two have the shapes of the shipped qft_n18 and qft_n63 circuits
    for Q, T in ((18, 138), (63, 498), (64, 500), (256, 500), (256, 2046), (512, 500), (1024, 500), (2048, 500), (4096, 500)):
        q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=Q)
        ir = CircuitIR.from_gate_tables(Q, q0, q1)
        dc = DeviceCircuit(ir, dev)
        B = max(64, (1 << 31) // (dc.ldq * T) // 2)  # about 1 GB of partitions
        for nx in (4, 8, 16):
            C = nx * nx
            g = torch.Generator(device=dev)
            g.manual_seed(Q + C)
            A = torch.randint(0, C, (B, T, dc.ldq), device=dev, generator=g, dtype=torch.int16).to(torch.uint8)
            # a partition drifts: most qubits keep their core from one row to the next
            keep = torch.rand((B, T, dc.ldq), device=dev, generator=g) < 0.9
            for t in range(1, T):
                A[:, t] = torch.where(keep[:, t], A[:, t - 1], A[:, t])
            del keep
            for wr, wname in (((2.0, 2.0), "integer"), ((1.0, 3.0), "real")):
                arch = multicore(nx, nx, ncm_qubits=4, ncp_qubits=max(2, 2 * Q // C), edge_weight_range=wr, network_type="mesh")
                dt = DistanceTable.from_network(arch.network, dev)
                for table in (("packed", "table") if wname == "integer" else ("packed",)):
                    out = torch.empty(B, dtype=torch.float64, device=dev)
                    for _ in range(2):
                        eng.cost_v3(dc, dt, A, out=out, table=table)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        eng.cost_v3(dc, dt, A, out=out, table=table)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 5
                    how = ("closed form" if table == "packed" and dt.kind == 1 and dt.grid.valid else
                           "u16 look-up" if dt.kind == 1 else "f64 pair counts" if C * C * 8 > 140 * 1024 and 512 < dc.ldq <= 16384 and dt.symmetric
                           else "f64 look-up (L2)" if C * C * 8 > 140 * 1024 else "f64 look-up")
                    print(f"{Q:5d} {T:5d} {nx:3d}x{nx:<3d} {wname:>8} {how:>22} {ms:8.3f} {B * T / ms / 1e6:10.2f} {B * T * dc.ldq / ms / 1e6:7.0f}", flush=True)
            del A
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
