"""Host logic that needs no GPU: config parsing, CSV schema, and the cross-rank reduction of the
driver exercised with two gloo ranks."""
import math
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from quantum_simulator_amd import driver

EXAMPLE = """\
mode = simulate
topology = mesh
num_cores_x = 2
num_cores_y = 1
ncm_qubits = 4
ncp_qubits = 130
edge_weight_range = 1.0,2.0
circuit_folders = circuits/gen
placers = RANDOM, BFS, TRHQA, HQA, GTRHQA
repeats = 3
out_csv = results.csv
plot_file = results_plot.png
"""


def test_parse_example_config(tmp_path):
    p = tmp_path / "cfg.txt"
    p.write_text(EXAMPLE + "# a comment\nseed = 7\n")
    cfg = driver.parse_config_file(str(p))
    assert cfg["topology"] == "mesh" and cfg["num_cores_x"] == 2 and cfg["ncp_qubits"] == 130
    assert cfg["edge_weight_range"] == (1.0, 2.0)
    assert cfg["circuit_folders"] == ["circuits/gen"]
    assert cfg["placers"] == ["RANDOM", "BFS", "TRHQA", "HQA", "GTRHQA"]
    assert cfg["repeats"] == 3 and cfg["seed"] == 7


@pytest.mark.parametrize("drop", ["mode", "topology", "ncp_qubits", "placers", "out_csv"])
def test_missing_key_is_an_assertion(tmp_path, drop):
    p = tmp_path / "cfg.txt"
    p.write_text("\n".join(l for l in EXAMPLE.splitlines() if not l.startswith(drop)))
    with pytest.raises(AssertionError):
        driver.parse_config_file(str(p))


def test_bad_values(tmp_path):
    p = tmp_path / "cfg.txt"
    p.write_text(EXAMPLE.replace("topology = mesh", "topology = ring"))
    with pytest.raises(AssertionError):
        driver.parse_config_file(str(p))
    p.write_text(EXAMPLE.replace("1.0,2.0", "5"))
    with pytest.raises(AssertionError):
        driver.parse_config_file(str(p))


def test_checks_come_in_the_reference_order(tmp_path, capsys):
    """config_parser.py:35-39 tests mode's value before it looks for topology: a config with a bad mode and no
    topology complains about the mode.  Values are read as config_parser.py:4-17 reads them: whatever int() and
    float() accept.  A circuit folder that does not exist is warned about, not refused (config_parser.py:70-72)."""
    p = tmp_path / "cfg.txt"
    lines = [l for l in EXAMPLE.splitlines() if not l.startswith("topology")]
    p.write_text("\n".join(lines).replace("mode = simulate", "mode = dream"))
    with pytest.raises(AssertionError, match="mode must be"):
        driver.parse_config_file(str(p))
    p.write_text("\n".join(lines))
    with pytest.raises(AssertionError, match="Missing required parameter: topology"):
        driver.parse_config_file(str(p))
    assert driver._scalar_or_list(" 1_000 ") == 1000 and driver._scalar_or_list("+7") == 7
    assert driver._scalar_or_list("1e3, -infinity") == [1000.0, float("-inf")]
    assert driver._scalar_or_list("2.5,") == 2.5 and driver._scalar_or_list("a, 2") == ["a", "2"]
    assert driver._scalar_or_list(",") == [] and driver._scalar_or_list("2.5") == "2.5"
    p.write_text(EXAMPLE.replace("circuits/gen", str(tmp_path / "nowhere")))
    cfg = driver.parse_config_file(str(p))
    assert cfg["circuit_folders"] == [str(tmp_path / "nowhere")]
    assert "Warning: circuit folder does not exist" in capsys.readouterr().out


def test_complete_topology_falls_back_to_4x4_all_to_all(capsys):
    # the parser accepts "complete", the topology builder does not: the driver's fall-back machine
    arch = driver.build_arch({"topology": "complete", "num_cores_x": 3, "num_cores_y": 3, "ncm_qubits": 4,
                              "ncp_qubits": 2, "edge_weight_range": (1.0, 2.0)})
    assert arch.network.number_of_nodes() == 16 and arch.network.number_of_edges() == 120
    assert arch.qpu_qubit_num == 16


def test_topologies_node_and_edge_order():
    from quantum_simulator_amd.multicore import build_topology
    g = build_topology("mesh", 2, 3)
    assert list(g.nodes) == list(range(6))
    assert list(g.edges()) == [(0, 3), (0, 1), (1, 4), (1, 2), (2, 5), (3, 4), (4, 5)]
    assert build_topology("tree", 2).number_of_nodes() == 7  # x is the height
    assert list(build_topology("star", 4).edges()) == [(0, 1), (0, 2), (0, 3)]
    assert build_topology("torus", 3, 3).number_of_edges() == 18
    assert list(build_topology("line", 3, 9).edges()) == [(0, 1), (1, 2)]
    with pytest.raises(ValueError):
        build_topology("complete", 2, 2)


def test_csv_schema(tmp_path):
    out = tmp_path / "r.csv"
    driver.write_csv(str(out), ["HQA", "BFS"], [{"file": "a.qasm", "HQA_time_mean": 1.0, "HQA_cost_mean": 386.0}])
    rows = out.read_text().splitlines()
    assert rows[0] == "file,HQA_time_mean,HQA_cost_mean,BFS_time_mean,BFS_cost_mean"
    assert rows[1] == "a.qasm,1.0,386.0,inf,inf"


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch.distributed as dist
    from quantum_simulator_amd import driver
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
    r = dist.get_rank()
    sums = np.array([[[10.0 * (r + 1), 1, 0.5, 1]], [[0.0, 0, 0.25 * r, r]]])
    mins = np.array([[10.0 * (r + 1)], [np.inf]])
    s, m = driver.reduce_table(sums, mins)
    assert s[0, 0].tolist() == [30.0, 2.0, 1.0, 2.0], s
    assert s[1, 0].tolist() == [0.0, 0.0, 0.25, 1.0], s
    assert m[0, 0] == 10.0 and np.isinf(m[1, 0])
    dist.destroy_process_group()
    print("rank", r, "ok")
""")


def test_reduce_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out
        assert "ok" in out


SHARD_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from quantum_simulator_amd import driver, placers

    class Fake:
        '''Deterministic stand-in: the cost depends on the circuit size and on how often this
        instance was asked, so a wrong shard or a double count shows up in the mean.'''
        def __init__(self, device=None):
            self.calls = 0
        def place_per_timeslice(self, circuit, arch):
            return [dict()]
        def per_timeslice_cost(self, circuit, part, arch):
            self.calls += 1
            if circuit.n_gates % 7 == 3 and self.calls == 1:
                raise KeyError("simulated failure")      # becomes inf for that repeat only
            return float(circuit.n_gates * 10 + circuit.n_qubits)

    placers.PLACERS.clear()
    placers.PLACERS.update({{"HQA": Fake, "TRHQA": Fake}})
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world)
    cfg = driver.parse_config_file({cfg!r})
    cfg["out_csv"] = {out!r} + os.environ.get("RANK", "0")
    res = driver.run_simulation(cfg, device="cpu", verbose=False)
    print("RESULT " + json.dumps([[r[k] for k in sorted(r) if k.endswith("cost_mean") or k.endswith("cost_min")] for r in res]))
    if world > 1:
        dist.destroy_process_group()
""")


def test_sharded_run_equals_single_rank(tmp_path):
    from quantum_simulator_amd import circuit_to_qasm_str, layered_circuit
    folder = tmp_path / "circ"
    folder.mkdir()
    for k, (q, t) in enumerate([(6, 4), (9, 3), (5, 7)]):
        (folder / f"c{k}.qasm").write_text(circuit_to_qasm_str(layered_circuit(q, t, 0.5, seed=k)))
    cfg = tmp_path / "cfg.txt"
    cfg.write_text(EXAMPLE.replace("circuits/gen", str(folder)).replace("RANDOM, BFS, TRHQA, HQA, GTRHQA", "HQA, TRHQA")
                   .replace("repeats = 3", "repeats = 5").replace("results_plot.png", "") + "seed = 3\n")
    script = tmp_path / "w.py"
    script.write_text(SHARD_WORKER.format(root=ROOT, cfg=str(cfg), out=str(tmp_path / "out.csv")))

    def run(world):
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29572")
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        outs = []
        for p in procs:
            out, _ = p.communicate(timeout=240)
            assert p.returncode == 0, out
            outs.append([l for l in out.splitlines() if l.startswith("RESULT ")][0])
        return outs

    single = run(1)[0]
    double = run(2)
    assert double[0] == double[1], "every rank must end with the full table"
    import json
    a, b = json.loads(single[7:]), json.loads(double[0][7:])
    assert len(a) == 3
    for ra, rb in zip(a, b):
        for x, y in zip(ra, rb):
            assert x == y or (math.isinf(x) and math.isinf(y))
    assert (tmp_path / "out.csv0").exists() and not (tmp_path / "out.csv1").exists()  # rank 0 writes the CSV


UNSEEDED_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from quantum_simulator_amd import driver, placers

    seen = []

    class Fake:
        def __init__(self, device=None):
            pass
        def place_per_timeslice(self, circuit, arch):
            seen.append([d["weight"].hex() for _, _, d in arch.network.edges(data=True)])
            return [dict()]
        def per_timeslice_cost(self, circuit, part, arch):
            return sum(d["weight"] for _, _, d in arch.network.edges(data=True))

    placers.PLACERS.clear()
    placers.PLACERS.update({{"HQA": Fake}})
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
    cfg = driver.parse_config_file({cfg!r})
    assert "seed" not in cfg
    cfg["out_csv"] = {out!r} + os.environ["RANK"]
    res = driver.run_simulation(cfg, device="cpu", verbose=False)
    print("RESULT " + json.dumps([seen[0], res[0]["HQA_cost_mean"], res[0]["HQA_cost_min"]]))
    dist.destroy_process_group()
""")


def test_two_ranks_without_a_seed_build_the_same_machine(tmp_path):
    """No ``seed`` in the config: rank 0 draws one and every rank builds its edge weights from it, so the
    reduced table describes one machine (mean == min when every repeat costs the same)."""
    import json
    from quantum_simulator_amd import circuit_to_qasm_str, layered_circuit
    folder = tmp_path / "circ"
    folder.mkdir()
    (folder / "c.qasm").write_text(circuit_to_qasm_str(layered_circuit(6, 4, 0.5, seed=1)))
    cfg = tmp_path / "cfg.txt"
    cfg.write_text(EXAMPLE.replace("circuits/gen", str(folder)).replace("RANDOM, BFS, TRHQA, HQA, GTRHQA", "HQA")
                   .replace("repeats = 3", "repeats = 4").replace("results_plot.png", "")
                   .replace("num_cores_x = 2", "num_cores_x = 3").replace("num_cores_y = 1", "num_cores_y = 3"))
    script = tmp_path / "w.py"
    script.write_text(UNSEEDED_WORKER.format(root=ROOT, cfg=str(cfg), out=str(tmp_path / "out.csv")))
    procs = []
    for r in range(2):
        # different hash seeds and start times: the ranks' own random streams differ
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29573",
                   PYTHONHASHSEED=str(r + 1))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out
        outs.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    assert outs[0][0] == outs[1][0], "edge weights differ between ranks"
    assert len(set(outs[0][0])) > 1
    assert outs[0][1:] == outs[1][1:] and outs[0][1] == outs[0][2]


def test_slice_shards_tile_the_circuit():
    """Contiguous pieces that cover every timeslice once; each piece's window starts one row early from slice 2 on."""
    from quantum_simulator_amd.driver import slice_shard
    for T in (0, 1, 2, 3, 7, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for rank in range(world):
                t0, t1, row0 = slice_shard(T, rank, world)
                assert 0 <= t0 <= t1 <= T
                assert row0 == (t0 - 1 if t0 >= 2 else t0)
                seen += list(range(t0, t1))
            assert seen == list(range(T))
