"""TEST INFRASTRUCTURE ONLY — a name adapter, not pytket.

The reference placers do ``from pytket import OpType`` at import time (Placer.py:2,
HQA_placer.py:3, multicore.py:4) and pytket is not installed.  Putting this directory ahead of the
reference on ``sys.path`` lets those files import unmodified; it re-exports this repository's own
circuit classes (file-order commands, qubits sorted by (register, index)) under the two names the
reference uses: ``pytket.OpType`` and ``pytket.qasm.circuit_from_qasm``.  Only
``oracle/gen_golden.py`` uses it, and only in the container that holds the reference.
"""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from quantum_simulator_amd.circuit import OpType, Qubit, Circuit  # noqa: E402,F401
