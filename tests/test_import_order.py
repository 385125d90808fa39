"""The shared library must end up on the same HIP runtime as torch whatever is used first."""
import subprocess
import sys

import pytest

from conftest import ROOT

CODE = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
from quantum_simulator_amd import engine
# a host helper first: this loads libqplace.so before the caller has imported torch
D = engine.apsp(4, [0, 1, 2, 3], [1, 2, 3, 0], np.array([1.0, 2.0, 3.0, 4.0]))
assert D[0, 2] == 3.0
assert "torch" in sys.modules, "loading the library must bring torch (and its HIP runtime) in first"
{gpu_part}
print("ok")
"""

GPU_PART = r"""
import torch
from quantum_simulator_amd import layered_gate_arrays
from quantum_simulator_amd.timeslices import CircuitIR
dev = torch.device("cuda", 0)
q0, q1, _ = layered_gate_arrays(40, 6, 0.5, seed=1)
ir = CircuitIR.from_gate_tables(40, q0, q1)
dc, dt, eng = engine.DeviceCircuit(ir, dev), engine.DistanceTable(D, dev), engine.Engine(dev)
A = torch.zeros((2, ir.T, dc.ldq), dtype=torch.uint8, device=dev)
assert eng.cost_v2(dc, dt, A, engine.QP_V1).cpu().tolist() == [0.0, 0.0]   # first device call: the sparse cost launcher
"""


def _run(gpu_part):
    r = subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, gpu_part=gpu_part)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_library_load_imports_torch_first():
    _run("")


@pytest.mark.gpu
def test_host_helper_before_torch_then_device_call():
    _run(GPU_PART)
