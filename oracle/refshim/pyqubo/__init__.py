"""TEST INFRASTRUCTURE ONLY — a name, not pyqubo.

Qubo_placer.py:5 does ``from pyqubo import Array, Binary`` at import time and pyqubo is not installed.
Its cost function (Qubo_placer.py:177-207) and its communication-count matrix (:208-245) never touch
pyqubo; they are what oracle/gen_golden_qubo_cost.py runs.  The two names exist so that the file
imports unmodified; using either raises, so nothing of pyqubo is stood in for and
``qubo_placer.place_per_timeslice`` stays unrunnable here.
"""


def _absent(*_a, **_k):
    raise ImportError("pyqubo is not installed: only qubo_placer's cost and count methods can run here")


Array = Binary = _absent
