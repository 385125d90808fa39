#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — writes tests/golden/broken_links.json.

Runs the *unmodified* reference ``simulate_broken_links`` of several placers (HQA_placer.py:325-361,
TRHQA_placer.py:628-669, BFS_placer.py:241-278, Random_placer.py:255-296) on golden
cases with a pinned seed of Python's global ``random`` and records, per simulation, which edges it broke
(as indices into ``G.edges()`` order) and the cost it reported (hex float; "inf" when infinite).  Only runs
where /root/reference exists; nothing from the reference is copied."""
import contextlib
import io
import json
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

RUNS = [("qft18_mesh4x4_u", "HQA", 4242, 6, 0.15), ("shor5_mesh2x2_i", "HQA", 99, 8, 0.3),
        ("qft29_torus8x8_u", "HQA", 5, 3, 0.1), ("qft18_mesh4x4_u", "TRHQA", 17, 5, 0.15),
        ("bigadder_mesh4x4_u", "BFS", 23, 5, 0.2)]  # Random_placer's variant calls an alternate placement: not covered
CLS = {"HQA": "hqa_placer", "TRHQA": "trhqa_placer", "BFS": "bfs_placer", "RANDOM": "random_placer"}


def main():
    R = G._load_reference()
    cases = {c[0]: c for c in G.CASES}
    out = {}
    for name, pname, seed, nsim, prob in RUNS:
        _, qf, topo, nxx, nyy, cap, wr, aseed = cases[name]
        circ = R["qasm"].circuit_from_qasm(os.path.join(G.REF, "circuits", qf), maxwidth=1024)
        arch = G._build_arch(R, topo, nxx, nyy, cap, wr, aseed)
        edges = list(arch.network.edges)
        eidx = {e: i for i, e in enumerate(edges)}
        placer = getattr(R[pname], CLS[pname])()
        random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            res = placer.simulate_broken_links(circ, arch, num_simulations=nsim, break_probability=prob)
        out[f"{name}/{pname}"] = dict(seed=seed, num_simulations=nsim, break_probability=prob,
                         sims=[dict(broken=sorted(eidx[e] for e in b),
                                    cost="inf" if c == float("inf") else float(c).hex()) for b, c in res])
        print(name, pname, [(len(s["broken"]), s["cost"]) for s in out[f"{name}/{pname}"]["sims"]], flush=True)
    with open(os.path.join(G.OUT, "broken_links.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
