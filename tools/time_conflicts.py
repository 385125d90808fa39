"""How much of a table look-up launch is LDS bank conflicts: the same launch on partitions whose look-ups all hit two
table entries (every qubit alternates between cores 0 and 1: the reads of an instruction share an address and are
broadcast) beside partitions that move every qubit to a random core: time_conflicts.py"""
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
    for Q, T, nx in ((18, 138, 4), (63, 498, 8), (256, 2046, 4), (1024, 500, 8)):
        q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=Q)
        ir = CircuitIR.from_gate_tables(Q, q0, q1)
        dc = DeviceCircuit(ir, dev)
        B = max(64, (1 << 29) // (dc.ldq * T))
        C = nx * nx
        g = torch.Generator(device=dev)
        g.manual_seed(Q)
        rand = torch.randint(0, C, (B, T, dc.ldq), device=dev, generator=g, dtype=torch.int16).to(torch.uint8)
        same = (torch.arange(T, device=dev) % 2).to(torch.uint8)[None, :, None].expand(B, T, dc.ldq).contiguous()
        for wr, wname in (((2.0, 2.0), "u16"), ((1.0, 3.0), "f64")):
            arch = multicore(nx, nx, ncm_qubits=4, ncp_qubits=max(2, 2 * Q // C), edge_weight_range=wr, network_type="mesh")
            dt = DistanceTable.from_network(arch.network, dev)
            for name, A in (("random cores", rand), ("two entries", same)):
                out = torch.empty(B, dtype=torch.float64, device=dev)
                for _ in range(2):
                    eng.cost_v3(dc, dt, A, out=out, table="table")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    eng.cost_v3(dc, dt, A, out=out, table="table")
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                print(f"{Q:5d} x {T:5d}  {C:3d} cores  {wname}  {name:13s} {ms:7.3f} ms  {B * T * dc.ldq / ms / 1e6:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
