"""Digest of what the annealer returns for fixed seeds (states and summed energy changes), and its time: for comparing two
builds of the library (QPLACE_LIB): anneal_digest.py [sweeps=60] [replicas=512]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from quantum_simulator_amd import layered_gate_arrays  # noqa: E402
from quantum_simulator_amd.engine import hops  # noqa: E402
from quantum_simulator_amd.multicore import build_topology, graph_tables  # noqa: E402
from quantum_simulator_amd.qubo import DeviceQubo, QuboModel  # noqa: E402
from quantum_simulator_amd.timeslices import CircuitIR  # noqa: E402

if __name__ == "__main__":
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dev = torch.device("cuda", 0)
    for Q, T, cap, topo, nx, ny in ((256, 8, 20, "mesh", 4, 4), (40, 5, 6, "line", 9, 1), (90, 6, 5, "torus", 5, 6)):
        q0, q1, _ = layered_gate_arrays(Q, T, 0.5, seed=4242)
        ir = CircuitIR.from_gate_tables(Q, q0, q1)
        nodes, eu, ev, _ = graph_tables(build_topology(topo, nx, ny))
        qm = QuboModel(ir, len(nodes), cap, hops(len(nodes), eu, ev))
        dq = DeviceQubo(qm, dev)
        dq.anneal(R, 1, seed=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        X, dE = dq.anneal(R, sweeps, seed=3)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        h = hashlib.sha1(X.cpu().numpy().tobytes() + dE.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"{topo}{nx}x{ny} Q={Q} T={T}: digest {h}  best dE {int(dE.min())}  {qm.N * R * sweeps / dt:.3e} attempts/s", flush=True)
