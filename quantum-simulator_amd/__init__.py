"""MI355X-native placement-cost evaluation engine behind the reference's Placer plug-in API.

Host side (this package): circuit front end, timeslice CSR, core topologies, Placer subclasses with
the reference's ``place_per_timeslice`` / ``per_timeslice_cost`` signatures.  Device side
(``csrc/``): hand-written gfx950 HIP kernels behind a C ABI (``include/qplace.h``), loaded with
ctypes from ``libqplace.so``.  There is no CPU fallback: calls that need the library raise if it is
missing.
"""
from .circuit import (Circuit, Command, Op, OpType, Qubit, circuit_from_qasm, circuit_from_qasm_str,
                      layered_circuit, layered_gate_arrays, circuit_to_qasm_str)

__all__ = ["Circuit", "Command", "Op", "OpType", "Qubit", "circuit_from_qasm",
           "circuit_from_qasm_str", "layered_circuit", "layered_gate_arrays", "circuit_to_qasm_str"]
