#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — writes tests/golden/qubo_cost.json.

For every golden case, rebuilds the circuit and architecture exactly as oracle/gen_golden.py did, turns stored
partitions back into the reference's list-of-dicts form and records what the *unmodified* reference
``qubo_placer`` returns for them:

* ``per_timeslice_cost`` (Qubo_placer.py:177-207): the gate-ordered cost on unweighted hop counts, an int
  (or the name of the exception it raises: a partition that leaves out a qubit it placed before, or an operand
  of a counted gate);
* ``intercorecommatrix`` (Qubo_placer.py:208-245): the symmetric core x core event counts.  The method returns
  nothing and draws a heat map; the matrix is taken from the ``imshow`` call it makes (matplotlib.pyplot.imshow
  is replaced by a recorder for the call, savefig by a no-op), stored as the upper triangle's non-zero
  entries.

Partitions: the sparse ones of BFS and RANDOM (only the slice's active qubits are keys), the dense one of HQA,
and one damaged copy of the BFS partition per case (a placed qubit removed from a later slice) for the
raising paths.  Qubo_placer.py imports pyqubo and neal at module level; oracle/refshim holds import-only names
for the two (neither method touches them).  Only runs where /root/reference exists; nothing of it is copied.
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

MAX_TERMS = 400_000  # cases whose walk is longer than this are skipped for the count matrix (it searches a list per event)


def _dicts(arr, allq, nodes):
    return [{q: nodes[int(c)] for q, c in zip(allq, row) if c != 0xFF} for row in arr]


def main():
    R = G._load_reference()
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import Qubo_placer  # the unmodified reference file
    placer = Qubo_placer.qubo_placer()
    out = {}
    for case in G.CASES:
        name, qf, topo, nxx, nyy, cap, wr, seed = case
        z = np.load(os.path.join(G.OUT, name + ".npz"))
        circ = R["qasm"].circuit_from_qasm(os.path.join(G.REF, "circuits", qf), maxwidth=1024)
        allq = list(circ.qubits)
        arch = G._build_arch(R, topo, nxx, nyy, cap, wr, seed)
        nodes = list(arch.network.nodes)
        n_visits = int(sum(len(cmd.qubits) for cmd in circ))
        rec = {}
        for pname in ("BFS", "RANDOM", "HQA", "BFS_damaged"):
            key = "part_" + pname.split("_")[0]
            if key not in z.files:
                continue
            arr = z[key].copy()
            damage = None
            if pname == "BFS_damaged":
                # remove one placed qubit from a slice in the middle of the circuit: the walk meets a qubit that is an
                # operand of a counted gate, or that it has seen on a core, and finds it on none
                rows = [t for t in range(1, arr.shape[0]) if (arr[t] != 0xFF).any() and (arr[:t] != 0xFF).any()]
                if not rows:
                    continue
                t = rows[len(rows) // 2]
                q = int(np.flatnonzero(arr[t] != 0xFF)[0])
                arr[t, q] = 0xFF
                damage = [t, q]
            part = _dicts(arr, allq, nodes)
            entry = {"damage": damage}
            with contextlib.redirect_stdout(io.StringIO()):
                try:
                    cost = placer.per_timeslice_cost(circ, part, arch)
                    entry["cost"] = int(cost)
                    assert cost == int(cost)
                except Exception as exc:  # noqa: BLE001 - the exception's class is the recorded result
                    entry["cost"] = None
                    entry["cost_raises"] = type(exc).__name__
            if n_visits <= MAX_TERMS:
                seen = []
                real_imshow, real_savefig = plt.imshow, plt.savefig
                plt.imshow = lambda m, *a, **k: (seen.append(np.array(m)), real_imshow(m, *a, **k))[1]
                plt.savefig = lambda *a, **k: None
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        placer.intercorecommatrix(circ, part, arch)
                    m = seen[0]
                    assert (m == m.T).all()
                    iu = np.argwhere(np.triu(m) != 0)
                    entry["comm_upper"] = [[int(a), int(b), int(m[a, b])] for a, b in iu]
                    entry["comm_total"] = int(m.sum())
                except Exception as exc:  # noqa: BLE001
                    entry["comm_upper"] = None
                    entry["comm_raises"] = type(exc).__name__
                finally:
                    plt.imshow, plt.savefig = real_imshow, real_savefig
                    plt.close("all")
            rec[pname] = entry
        out[name] = rec
        print(name, {k: (v.get("cost"), v.get("cost_raises"), v.get("comm_total"), v.get("comm_raises")) for k, v in rec.items()},
              flush=True)
    with open(os.path.join(G.OUT, "qubo_cost.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
