"""fp64 table of 256 cores (too large for LDS) on short rows, partitions of a few KB to a few tens of KB: where pair
counts (a pass over all counters per item) overtake gathers from device memory: time_small_parts.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from quantum_simulator_amd import layered_gate_arrays  # noqa: E402
from quantum_simulator_amd.engine import DeviceCircuit, DistanceTable, Engine  # noqa: E402
from quantum_simulator_amd.multicore import multicore  # noqa: E402
from quantum_simulator_amd.timeslices import CircuitIR  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    eng = Engine(dev)
    arch = multicore(16, 16, ncm_qubits=4, ncp_qubits=4, edge_weight_range=(1.0, 3.0), network_type="mesh")
    dt = DistanceTable.from_network(arch.network, dev)
    for Q, T in ((18, 138), (29, 226), (63, 250), (63, 498), (63, 1000), (128, 300), (256, 120)):
        q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=Q)
        ir = CircuitIR.from_gate_tables(Q, q0, q1)
        dc = DeviceCircuit(ir, dev)
        B = max(64, (1 << 29) // (dc.ldq * ir.T))
        g = torch.Generator(device=dev)
        g.manual_seed(Q)
        A = torch.randint(0, 256, (B, ir.T, dc.ldq), device=dev, generator=g, dtype=torch.int16).to(torch.uint8)
        keep = torch.rand((B, ir.T, dc.ldq), device=dev, generator=g) < 0.9
        for t in range(1, ir.T):
            A[:, t] = torch.where(keep[:, t], A[:, t - 1], A[:, t])
        del keep
        out = torch.empty(B, dtype=torch.float64, device=dev)
        for _ in range(3):
            eng.cost_v3(dc, dt, A, out=out)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.cost_v3(dc, dt, A, out=out)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / 10
        print(f"Q={Q:4d} T={ir.T:5d} partition {ir.T * dc.ldq / 1024:6.1f} KB  B={B:7d}  {ms:7.3f} ms  {B * ir.T * dc.ldq / ms / 1e6:7.0f} GB/s  sum {float(out.sum()):.6e}", flush=True)
        del A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
