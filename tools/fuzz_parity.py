"""Randomised differential run of the device kernels against the CPU oracle: draws shapes the fixed
parametrisations of tests/ do not list and calls the same test bodies with them.
fuzz_parity.py [seconds=120] [seed=0]   (run on a GPU box from the repository root; exits non-zero on a mismatch)"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_cost as TC  # noqa: E402
import test_gpu_placers as TP  # noqa: E402
import test_gpu_qubo as TQ  # noqa: E402
from quantum_simulator_amd import engine  # noqa: E402


def unwrap(f):
    return getattr(f, "__wrapped__", f)


def cost_matrix_case(gpu, Q, T, C, cap, kind, seed):
    """Every op x core matrix the oracle's placement loop builds on a random circuit, bit for bit (the body of
    tests/test_gpu_cost.py::test_cost_matrix_bits on a circuit drawn here instead of a fixture)."""
    from oracle import placer_oracle as O
    from quantum_simulator_amd import layered_gate_arrays
    from quantum_simulator_amd.timeslices import CircuitIR
    _, eng_mod, eng, dev = gpu
    rng = np.random.default_rng(seed)
    q0, q1, _ = layered_gate_arrays(Q, T, float(rng.choice([0.3, 0.6, 0.9])), seed=seed)
    keep = rng.random(len(q0)) > 0.2
    ir = CircuitIR.from_gate_tables(Q, q0[keep], q1[keep])
    eu = list(range(C)) + [int(x) for x in rng.integers(0, C, size=C // 2)]
    ev = [(i + 1) % C for i in range(C)] + [int(x) for x in rng.integers(0, C, size=C // 2)]
    ok = [k for k in range(len(eu)) if eu[k] != ev[k]]
    eu, ev = [eu[k] for k in ok], [ev[k] for k in ok]
    D = eng_mod.apsp(C, eu, ev, rng.uniform(1, 3, size=len(eu)))
    dc, dt = eng_mod.DeviceCircuit(ir, dev), eng_mod.DistanceTable(D, dev)
    rec, Dl = [], D.tolist()
    try:
        if kind == "gtrhqa":
            O.place_gtrhqa(ir.slices, ir.gate_q, ir.gate_mr, Q, C, cap, Dl, rec)
        else:
            O.place_hqa_family(ir.slices, ir.gate_q, ir.gate_mr, Q, C, cap, Dl, kind, rec)
    except Exception:  # the reference gives up on this instance (no room): matrices up to there still count
        pass
    rec = rec[:300]
    if not rec:
        return
    B = len(rec)
    max_ops = max(len(r[1]) for r in rec)
    t = np.array([r[0] for r in rec], np.int32)
    a = np.zeros((B, dc.ldq), np.uint8)
    ops = np.zeros((B, max_ops), np.int32)
    n_ops = np.zeros(B, np.int32)
    avail = np.zeros((B, C), np.uint8)
    n_avail = np.zeros(B, np.int32)
    for b, (tt, rops, rav, M, live) in enumerate(rec):
        a[b, :Q] = live
        ops[b, :len(rops)] = [ir.edge_of_gate[g] for g, _, _ in rops]
        n_ops[b] = len(rops)
        avail[b, :len(rav)] = rav
        n_avail[b] = len(rav)
    code = {"hqa": eng_mod.QP_CM_HQA, "trhqa": eng_mod.QP_CM_TRHQA, "gtrhqa": eng_mod.QP_CM_GTRHQA}[kind]
    out = eng.cost_matrix(code, dc, dt, *[torch.from_numpy(x).to(dev) for x in (t, a, ops, n_ops, avail, n_avail)]).cpu().numpy()
    for b, (tt, rops, rav, M, live) in enumerate(rec):
        want = np.array(M, np.float64).reshape(len(rops), len(rav))
        got = out[b, :len(rops), :len(rav)]
        assert got.tobytes() == want.tobytes(), (kind, b, tt)


def hqa_batch_case(Q, T, nx, ny, topo, cap, kind, B, seed):
    """A native HQA / TRHQA placement batch (qp_hqa_place_batch) from shuffled starting assignments against the
    sequential restatement, row for row; with S > 1 damaged copies of the network, one per instance."""
    import random
    from oracle import placer_oracle as O
    from quantum_simulator_amd import layered_circuit
    from quantum_simulator_amd.multicore import multicore
    from quantum_simulator_amd.placers import PLACERS
    rng = np.random.default_rng(seed)
    circ = layered_circuit(Q, T, float(rng.choice([0.3, 0.6, 0.9])), seed=seed % 100000)
    random.seed(seed)
    wr = (2.0, 2.0) if seed % 3 == 0 else (1.0, 3.0)
    arch = multicore(nx, ny, ncp_qubits=cap, edge_weight_range=wr, network_type=topo)
    placer = PLACERS[kind.upper()]()
    ctx = placer._context(circ, arch)
    C = ctx.arch.C
    if Q > C * cap or cap < 2:
        return
    slots = np.repeat(np.arange(C), cap)
    inits = np.stack([rng.permutation(slots)[:Q] for _ in range(B)]).astype(np.uint8)
    ir = ctx.ir
    if seed % 2:
        got = placer.place_batch(circ, arch, initial=inits, threads=int(rng.integers(1, 5)))
        tables = [ctx.arch.D] * B
    else:  # every instance on its own damaged network
        W = np.tile(np.asarray(ctx.arch.w, np.float64), (B, 1))
        W[rng.random(W.shape) < 0.1] = np.inf
        W[0] = ctx.arch.w
        tabs = engine.apsp_batch(C, ctx.arch.eu, ctx.arch.ev, W)
        got = placer.place_batch(circ, arch, initial=inits, tables=tabs, tab_index=np.arange(B, dtype=np.int32))
        tables = list(tabs)
    for b in range(B):
        want = O.place_hqa_family(ir.slices, ir.gate_q, ir.gate_mr, Q, C, cap, tables[b].tolist(), kind, init=inits[b].tolist())
        assert (got[b] == np.array(want, np.uint8)).all(), (kind, b, np.argwhere(got[b] != np.array(want, np.uint8))[:4])


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda", 0)
    gpu = (torch, engine, engine.Engine(dev), dev)
    t0 = time.time()
    runs, fails = {}, 0
    last = [t0]

    def pick(lo, hi):
        return int(rng.integers(lo, hi + 1))

    def wide_q():  # favour the row-length classes the launchers switch on (<= 512, 513..1024, > 1024)
        return [pick(2, 80), pick(81, 512), pick(513, 1024), pick(1025, 1600)][pick(0, 3)]

    def draw():
        kind = pick(0, 14)
        if os.environ.get("QP_FUZZ_KIND"):  # one kind of case only
            kind = int(os.environ["QP_FUZZ_KIND"])
        if kind == 14:  # rows of at most 512 bytes, 64 partitions or more: four partitions per wave step, runs of steps or of quads
            Q = pick(2, 512)
            ldq = (Q + 15) // 16 * 16
            T = pick(1, max(1, min(9000, 300000 // ldq)))
            integer = bool(pick(0, 1))
            C = pick(2, 256) if integer else pick(2, 128)
            B = pick(64, 200) if T * ldq > 60000 else pick(64, 3000)
            return "cost_v3_short_rows_quads", TC.test_cost_v3_short_rows_four_partitions, (gpu, Q, T, C, B, integer, float(rng.choice([0.02, 0.3, 1.0])))
        if kind == 13:  # communication counts from the pair counters of the row walk: whole partitions or chunks
            B = pick(50, 63) if pick(0, 2) == 0 else pick(64, 500)
            T = pick(16384 // B + 1, 16384 // B + 60)
            return "comm_matrix_row_walk", TP.test_comm_matrix_row_walk, (pick(513, 1024), T, pick(2, 255), B)
        if kind == 12:  # gate-ordered cost as a row walk: whole partitions or chunks of slices, looked up or pair-counted
            if pick(0, 1):  # rows of at most 512 bytes: several partitions per wave step, from 64 MB on
                Q = pick(8, 512)
                T = pick(20, 250)
                B = (1 << 26) // ((Q + 15) // 16 * 16) // T + pick(1, 100)
                C = pick(2, 255)
                integer = bool(pick(0, 1)) or C > 127
                return "cost_v2_row_walk", TC.test_cost_v2_row_walk, (gpu, Q, T, C, B, float(rng.choice([0.0, 0.1, 0.3])), integer)
            if pick(0, 1):
                B, T = pick(1024, 1400), pick(1, 40)
            else:
                B = pick(270, 600)
                T = pick(65536 // B - 20, min(250, 65536 // B + 120))
            return "cost_v2_row_walk", TC.test_cost_v2_row_walk, (gpu, pick(513, 1024), T, pick(2, 255), B, float(rng.choice([0.0, 0.1, 0.3])),
                                                                  bool(pick(0, 1)))
        if kind == 11:  # rows of at most 512 bytes through the grouped gate list: every table form, one or several items
            Q = pick(2, 512)
            ldq = (Q + 15) // 16 * 16
            T = pick(1, max(1, min(9000, 400000 // ldq)))
            integer = bool(pick(0, 1))
            return "cost_v3_short_rows", TC.test_cost_v3_short_rows_in_groups, (gpu, Q, T, pick(2, 256), pick(1, 40), integer,
                                                                                float(rng.choice([0.02, 0.3, 1.0])))
        if kind == 10:  # long rows, table in LDS, 64 partitions or more: four partitions per wave step
            integer = bool(pick(0, 1))
            return "cost_v3_four_partitions", TC.test_cost_v3_four_partitions_per_wave, (gpu, pick(513, 1024), pick(1, 80), pick(2, 256) if integer else pick(2, 128),
                                                                                         pick(64, 140), integer)
        if kind == 9:
            C = [4, 9, 16, 25][pick(0, 3)]
            return "qubo_energy_fields", TQ.test_energy_and_fields_exact, ((torch, dev), pick(8, 70), pick(1, 7), C, pick(2, 12), pick(0, 1 << 20))
        if kind == 8:
            topo, nx, ny = [("mesh", pick(2, 8), pick(1, 8)), ("torus", pick(3, 8), pick(3, 8)), ("line", pick(2, 40), 1),
                            ("star", pick(3, 30), 1), ("all_to_all", pick(2, 4), pick(1, 4))][pick(0, 4)]
            return "hqa_place_batch", hqa_batch_case, (pick(2, 120), pick(2, 25), nx, ny, topo, pick(2, 6), ["hqa", "trhqa"][pick(0, 1)],
                                                       pick(1, 7), pick(0, 1 << 30))
        if kind == 7:
            return "cost_v3_multi", TP.test_cost_v3_with_a_table_per_partition, (wide_q(), pick(1, 40), pick(2, 256), pick(1, 6))
        if kind == 6:  # rows longer than 1 KB: one or four 16-byte pieces per thread, one or several chunks of rows
            Q = pick(16385, 24000) if pick(0, 5) == 0 else pick(1025, 6000)
            T = pick(1, max(1, min(600, 1200000 // Q)))
            topo, nx, ny = [("mesh", pick(1, 16), pick(2, 16)), ("torus", pick(3, 16), pick(3, 16)), ("line", pick(2, 200), 1),  # (two cores or more: one core has no distances)
                            ("star", pick(3, 60), 1), ("all_to_all", pick(2, 6), pick(1, 6))][pick(0, 4)]
            wt = float(pick(1, 5)) if pick(0, 1) else 0
            return "cost_v3_wide_rows", TC.test_cost_v3_wide_rows, (gpu, Q, T, topo, nx, ny, wt, pick(1, 3), float(rng.choice([0.02, 0.3, 1.0])))
        if kind == 5:
            C = pick(2, 256)
            cap = pick(2, 8)
            Q = min(pick(4, 400), C * cap)
            return "cost_matrix", cost_matrix_case, (gpu, Q, pick(2, 10), C, cap, ["hqa", "trhqa", "gtrhqa"][pick(0, 2)], pick(0, 1 << 30))
        if kind == 0:
            return "cost_v3_random", TC.test_cost_v3_random, (gpu, wide_q(), pick(1, 45), pick(2, 256), pick(1, 6), bool(pick(0, 1)))
        if kind == 1:
            return "cost_v2_random", TC.test_cost_v2_random, (gpu, wide_q(), pick(1, 40), pick(2, 255), pick(1, 9), bool(pick(0, 1)))
        if kind == 2:
            return "cost_v3_pair_counts", TC.test_cost_v3_pair_counts, (gpu, pick(513, 1024), pick(1, 140), pick(136, 256), pick(1, 5),
                                                                         bool(pick(0, 3) == 0), float(rng.choice([0.02, 0.3, 1.0])))
        if kind == 3:
            C = pick(2, 256)
            cap = pick(2, 10)
            Q = min(pick(2, 1024), C * cap)
            return "gtrhqa_place", TP.test_gtrhqa_device_placement_matches_oracle, (Q, pick(1, 14), C, cap, pick(1, 4), bool(pick(0, 4) == 0 and C >= 6), bool(pick(0, 1)))
        return "core_activity", TP.test_core_activity_and_comm_matrix_random, (wide_q() if pick(0, 1) else pick(2, 300), pick(1, 60), pick(2, 256), pick(1, 4))

    while time.time() - t0 < budget:
        name, fn, args = draw()
        shown = args[1:] if args and args[0] is gpu else args
        try:
            c0 = time.time()
            unwrap(fn)(*args)
            runs[name] = runs.get(name, 0) + 1
            if time.time() - last[0] > 30:  # a line every half minute: a silent run is taken for a hung one
                last[0] = time.time()
                print(f"... {sum(runs.values())} cases after {time.time() - t0:.0f} s", flush=True)
            if os.environ.get("QP_FUZZ_VERBOSE"):
                print(f"ok {name}{shown} {time.time() - c0:.3f} s", flush=True)
        except BaseException as e:  # AssertionError and anything a launch raises
            if isinstance(e, KeyboardInterrupt):
                raise
            fails += 1
            print(f"FAIL {name}{shown}: {type(e).__name__}: {str(e)[:300]}", flush=True)
            traceback.print_exc(limit=3)
    print(f"{sum(runs.values())} cases passed in {time.time() - t0:.0f} s: {runs}; {fails} failed", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
