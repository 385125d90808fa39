#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — writes tests/golden/activity.json.

For every golden case, rebuilds the circuit and architecture exactly as oracle/gen_golden.py did, turns
the stored HQA / TRHQA / GTRHQA partitions back into the reference's list-of-dicts form and records what
the *unmodified* reference returns for ``per_core_intercore_activity`` (HQA_placer.py:288-321, same body
in TRHQA_placer.py:586-619) as a list of counts in core-index order.  Only runs where /root/reference
exists; nothing from the reference is copied.
"""
import contextlib
import io
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

import numpy as np  # noqa: E402


def main():
    R = G._load_reference()
    out = {}
    for case in G.CASES:
        name, qf, topo, nxx, nyy, cap, wr, seed = case
        z = np.load(os.path.join(G.OUT, name + ".npz"))
        circ = R["qasm"].circuit_from_qasm(os.path.join(G.REF, "circuits", qf), maxwidth=1024)
        allq = list(circ.qubits)
        arch = G._build_arch(R, topo, nxx, nyy, cap, wr, seed)
        nodes = list(arch.network.nodes)
        placer = R["HQA"].hqa_placer()
        rec = {}
        for pname in ("HQA", "TRHQA", "GTRHQA"):
            key = f"part_{pname}"
            if key not in z.files:
                continue
            arr = z[key]
            part = [{q: nodes[int(c)] for q, c in zip(allq, row)} for row in arr]
            with contextlib.redirect_stdout(io.StringIO()):
                got = placer.per_core_intercore_activity(circ, part, arch)
            rec[pname] = [int(got[n]) for n in nodes]
        out[name] = rec
        print(name, {k: sum(v) for k, v in rec.items()}, flush=True)
    with open(os.path.join(G.OUT, "activity.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
