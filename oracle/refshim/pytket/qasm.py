"""``pytket.qasm`` name adapter (see the package docstring)."""
from quantum_simulator_amd.circuit import circuit_from_qasm, circuit_from_qasm_str  # noqa: F401
