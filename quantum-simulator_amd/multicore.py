"""Core-graph model with the reference's constructor and attribute names (multicore.py:122-148).

``multicore(num_qpus_x, num_qpus_y, ncm_qubits, ncp_qubits, edge_weight_range, network_type)`` builds
``.network`` (a networkx graph whose nodes are the integers 0..C-1, edge attribute ``weight``, node
attribute ``qpu``), ``.qpus`` and ``.qpu_qubit_num``.  The graph comes from the same networkx
generators the reference calls (multicore.py:6-56) so node order and ``G.edges()`` order are the
same, and each edge weight is one ``random.uniform(lo, hi)`` draw from the global ``random`` state
in ``G.edges()`` order (multicore.py:159-161) — seeding ``random`` the same way gives the same
weights.  ``ncm_qubits`` is stored and, as in the reference, enters no cost.
"""
import random

import networkx as nx
import numpy as np

TOPOLOGIES = ("line", "star", "tree", "mesh", "torus", "all_to_all")


def build_topology(topology, num_qpus_x, num_qpus_y=None):
    kind = str(topology).lower()
    if kind in ("mesh", "torus"):
        if num_qpus_y is None:
            raise ValueError(f"{kind} requires num_qpus_y")
        grid = nx.grid_2d_graph(num_qpus_x, num_qpus_y, periodic=(kind == "torus"))
        return nx.convert_node_labels_to_integers(grid)  # label of (i, j) = i * ny + j
    if kind == "line":
        return nx.path_graph(num_qpus_x)
    if kind == "star":
        if num_qpus_x < 1:
            raise ValueError("star requires at least 1 node")
        return nx.star_graph(num_qpus_x - 1)
    if kind == "tree":
        return nx.balanced_tree(r=2, h=num_qpus_x)  # x is the height: 2**(h+1) - 1 cores
    if kind == "all_to_all":
        return nx.complete_graph(num_qpus_x if num_qpus_y is None else num_qpus_x * num_qpus_y)
    raise ValueError(f"Unknown topology: {topology}")


class qpu:
    """One core.  Only ``available_qubits`` is read by placers (BFS_placer.py:106-107)."""

    def __init__(self, id, ncm_qubits, ncp_qubits):
        self.qpuid = id
        self.ncm_qubits = ncm_qubits
        self.ncp_qubits = ncp_qubits
        self.available_qubits = ncp_qubits


class multicore:
    def __init__(self, num_qpus_x, num_qpus_y, ncm_qubits=4, ncp_qubits=16, edge_weight_range=(1, 10),
                 network_type="all_to_all"):
        self.network = build_topology(network_type, num_qpus_x, num_qpus_y)
        lo, hi = edge_weight_range
        for u, v in self.network.edges():
            self.network[u][v]["weight"] = random.uniform(lo, hi)
        self.qpus = []
        for k, node in enumerate(self.network.nodes):
            core = qpu(k, ncm_qubits, ncp_qubits)
            self.network.nodes[node]["type"] = "qpu"
            self.network.nodes[node]["qpu"] = core
            self.qpus.append(core)
        self.qpu_qubit_num = ncp_qubits
        self.ncm_qubits = ncm_qubits
        self.network_type = network_type

    def get_edge_weights_summary(self):
        w = [d["weight"] for _, _, d in self.network.edges(data=True)]
        return {"min": min(w), "max": max(w), "mean": float(np.mean(w)), "std": float(np.std(w))}


def graph_tables(network):
    """(nodes, eu, ev, w): node list and the edge list as index arrays, in ``G.edges()`` order."""
    nodes = list(network.nodes)
    index = {n: i for i, n in enumerate(nodes)}
    eu, ev, w = [], [], []
    for a, b, d in network.edges(data=True):
        eu.append(index[a])
        ev.append(index[b])
        w.append(float(d.get("weight", 1)))
    return nodes, np.asarray(eu, np.int32), np.asarray(ev, np.int32), np.asarray(w, np.float64)
