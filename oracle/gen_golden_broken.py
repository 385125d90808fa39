#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — writes tests/golden/broken_links.json.

Runs the *unmodified* reference ``hqa_placer.simulate_broken_links`` (HQA_placer.py:325-361) on golden
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

RUNS = [("qft18_mesh4x4_u", 4242, 6, 0.15), ("shor5_mesh2x2_i", 99, 8, 0.3), ("qft29_torus8x8_u", 5, 3, 0.1)]


def main():
    R = G._load_reference()
    cases = {c[0]: c for c in G.CASES}
    out = {}
    for name, seed, nsim, prob in RUNS:
        _, qf, topo, nxx, nyy, cap, wr, aseed = cases[name]
        circ = R["qasm"].circuit_from_qasm(os.path.join(G.REF, "circuits", qf), maxwidth=1024)
        arch = G._build_arch(R, topo, nxx, nyy, cap, wr, aseed)
        edges = list(arch.network.edges)
        eidx = {e: i for i, e in enumerate(edges)}
        placer = R["HQA"].hqa_placer()
        random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            res = placer.simulate_broken_links(circ, arch, num_simulations=nsim, break_probability=prob)
        out[name] = dict(seed=seed, num_simulations=nsim, break_probability=prob,
                         sims=[dict(broken=sorted(eidx[e] for e in b),
                                    cost="inf" if c == float("inf") else float(c).hex()) for b, c in res])
        print(name, [(len(s["broken"]), s["cost"]) for s in out[name]["sims"]], flush=True)
    with open(os.path.join(G.OUT, "broken_links.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
