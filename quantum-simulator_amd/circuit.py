"""Circuit front end: qubits, commands, an OpenQASM 2.0 reader and a seeded synthetic generator.

The reference reads circuits with ``pytket.qasm.circuit_from_qasm(path, maxwidth=1024)``
(main.py:80) and its placers only touch this much of the resulting object:

* iteration over the circuit yields commands (``list(circuit)``, Placer.py:21);
* ``command.qubits`` is a list of hashable, ``==``-comparable qubit objects
  (Placer.py:29, HQA_placer.py:66-72) and ``command.op.type`` is compared with
  ``OpType.Measure`` / ``OpType.Reset`` (HQA_placer.py:67);
* ``circuit.qubits`` lists every qubit once, ordered by (register name, index)
  (HQA_placer.py:32, the order the initial block assignment walks, HQA_placer.py:40-44);
* ``str(qubit)`` is ``"reg[i]"`` (sort key of Qubo_placer.py:48).

The classes here offer exactly that surface, so the placers in this package accept either these
objects or real pytket circuits.  Commands are yielded in QASM file order.
"""
from __future__ import annotations

import re
import zlib
from typing import Iterable, List, Sequence, Tuple

import numpy as np


class OpType:
    """Gate kinds.  Only Measure and Reset are ever tested by the placers."""

    class _Kind:
        __slots__ = ("name",)

        def __init__(self, name):
            self.name = name

        def __repr__(self):
            return "OpType." + self.name

    Measure = _Kind("Measure")
    Reset = _Kind("Reset")
    Barrier = _Kind("Barrier")
    Conditional = _Kind("Conditional")
    CustomGate = _Kind("CustomGate")
    _named = {}

    @classmethod
    def of(cls, qasm_name: str):
        k = cls._named.get(qasm_name)
        if k is None:
            k = cls._Kind(_PYTKET_NAMES.get(qasm_name, qasm_name))
            cls._named[qasm_name] = k
            setattr(cls, k.name, k)
        return k


_PYTKET_NAMES = {
    "cx": "CX", "cz": "CZ", "swap": "SWAP", "rz": "Rz", "rx": "Rx", "ry": "Ry", "sx": "SX",
    "x": "X", "y": "Y", "z": "Z", "h": "H", "s": "S", "sdg": "Sdg", "t": "T", "tdg": "Tdg",
    "ccx": "CCX", "cy": "CY", "ch": "CH", "crz": "CRz", "cu1": "CU1", "cu3": "CU3", "u1": "U1",
    "u2": "U2", "u3": "U3", "id": "noop", "sxdg": "SXdg", "cswap": "CSWAP", "rzz": "ZZPhase",
}
for _n in _PYTKET_NAMES:
    OpType.of(_n)


class Qubit:
    """A qubit named ``reg[index]``.

    The hash is a fixed function of (register name, index) so that ``list(set(qubits))``, which the
    BFS and Random placers iterate (BFS_placer.py:112-115, Random_placer.py:41-45), has the same
    order in every process regardless of PYTHONHASHSEED.
    """

    __slots__ = ("reg", "index", "_h")

    def __init__(self, reg: str, index: int):
        self.reg = reg
        self.index = int(index)
        self._h = hash((zlib.crc32(reg.encode()), self.index))

    def __hash__(self):
        return self._h

    def __eq__(self, other):
        return isinstance(other, Qubit) and self.index == other.index and self.reg == other.reg

    def __lt__(self, other):
        return (self.reg, self.index) < (other.reg, other.index)

    def __repr__(self):
        return f"{self.reg}[{self.index}]"

    __str__ = __repr__


class Op:
    __slots__ = ("type", "name", "params")

    def __init__(self, type_, name, params=""):
        self.type = type_
        self.name = name
        self.params = params


class Command:
    __slots__ = ("op", "qubits", "bits")

    def __init__(self, op: Op, qubits: List[Qubit], bits: Sequence[Tuple[str, int]] = ()):
        self.op = op
        self.qubits = qubits
        self.bits = list(bits)

    def __repr__(self):
        return f"{self.op.name} {', '.join(map(str, self.qubits))}"


class _GateRun:
    """A run of plain gate statements kept as arrays (what the native scanner returns): name id per gate,
    the distinct names and their kinds, parameter text spans, qubit indices in declaration order (-1 = unused
    slot).  Command objects are only made when somebody iterates the circuit."""
    __slots__ = ("text", "name_id", "names", "kinds", "param_pos", "qidx")

    def __init__(self, text, name_id, names, kinds, param_pos, qidx):
        self.text, self.name_id, self.names, self.kinds = text, name_id, names, kinds
        self.param_pos, self.qidx = param_pos, qidx

    def __len__(self):
        return len(self.name_id)

    def commands(self, qubit_of):
        ops_plain = [Op(k, n, "") for n, k in zip(self.names, self.kinds)]
        out = []
        text = self.text
        for nid, (pb, pl), qs in zip(self.name_id.tolist(), self.param_pos.tolist(), self.qidx.tolist()):
            op = ops_plain[nid] if pb < 0 else Op(self.kinds[nid], self.names[nid], text[pb:pb + pl])
            out.append(Command(op, [qubit_of[q] for q in qs if q >= 0]))
        return out


class Circuit:
    """Ordered list of commands over named qubit registers."""

    def __init__(self):
        self._items = []          # Command objects and _GateRun blocks, in program order
        self._count = 0
        self._flat = None         # all commands as objects, made on first iteration
        self._qregs: List[Tuple[str, int]] = []
        self._qubit_objs = {}
        self._decl_order = []     # qubit objects in declaration order (what _GateRun indices refer to)

    def add_q_register(self, name: str, size: int):
        self._qregs.append((name, int(size)))
        for i in range(int(size)):
            q = Qubit(name, i)
            self._qubit_objs[(name, i)] = q
            self._decl_order.append(q)

    def qubit(self, reg: str, index: int) -> Qubit:
        return self._qubit_objs[(reg, index)]

    def add_command(self, name: str, qubits: Sequence[Qubit], params: str = "", optype=None, bits=()):
        t = optype if optype is not None else OpType.of(name)
        self._items.append(Command(Op(t, name, params), list(qubits), bits))
        self._count += 1
        self._flat = None

    def add_run(self, run: "_GateRun"):
        if len(run):
            self._items.append(run)
            self._count += len(run)
            self._flat = None

    def _commands_list(self) -> List[Command]:
        if self._flat is None:
            flat = []
            for it in self._items:
                if isinstance(it, _GateRun):
                    flat.extend(it.commands(self._decl_order))
                else:
                    flat.append(it)
            self._flat = flat
        return self._flat

    @property
    def qubits(self) -> List[Qubit]:
        return sorted(self._qubit_objs.values())

    @property
    def n_qubits(self) -> int:
        return len(self._qubit_objs)

    @property
    def n_gates(self) -> int:
        return self._count

    def __iter__(self):
        return iter(self._commands_list())

    def __len__(self):
        return self._count

    def get_commands(self):
        return list(self._commands_list())

    def gate_tables(self):
        """(qubit index table int32 [n_gates, k] in ``self.qubits`` order with -1 for unused slots, bool
        [n_gates] "is Measure or Reset") without making command objects for array-backed runs."""
        order = {q: i for i, q in enumerate(self.qubits)}
        decl_to_sorted = np.array([order[q] for q in self._decl_order] + [-1], np.int32)  # [-1] maps -1 to -1
        width = 1
        for it in self._items:
            width = max(width, it.qidx.shape[1] if isinstance(it, _GateRun) else len(it.qubits))
        tabs, mrs = [], []
        for it in self._items:
            if isinstance(it, _GateRun):
                q = decl_to_sorted[it.qidx]
                if q.shape[1] < width:
                    q = np.concatenate([q, np.full((len(q), width - q.shape[1]), -1, np.int32)], axis=1)
                tabs.append(q)
                kind_mr = np.array([k in (OpType.Measure, OpType.Reset) for k in it.kinds], bool)
                mrs.append(kind_mr[it.name_id])
            else:
                row = np.full((1, width), -1, np.int32)
                row[0, :len(it.qubits)] = [order[q] for q in it.qubits]
                tabs.append(row)
                mrs.append(np.array([it.op.type in (OpType.Measure, OpType.Reset)], bool))
        if not tabs:
            return np.zeros((0, width), np.int32), np.zeros(0, bool)
        return np.concatenate(tabs), np.concatenate(mrs)


# ----------------------------------------------------------------------------------------------
# OpenQASM 2.0 reader
# ----------------------------------------------------------------------------------------------

_ARG_RE = re.compile(r"^([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\s*(\d+)\s*\])?$")
_GATE_RE = re.compile(r"^([A-Za-z_][A-Za-z0-9_]*)\s*(\((.*)\))?\s*(.*)$", re.S)


class QasmError(ValueError):
    pass


def _strip_comments(text: str) -> str:
    return re.sub(r"//[^\n]*", "", text)


class _Statements:
    """Cursor over the ';'-terminated statements of a program text: ``next(pos)`` returns (statement with
    comments removed, position after it), or (None, len) at the end.  A ``gate ... { ... }`` definition
    is one statement.  The position of the next comment is remembered so a file is scanned once."""

    def __init__(self, text: str):
        self.text = text
        self._comment = -2  # position of the first "//" at or after the last query, -1: none left

    def _next_comment(self, k: int) -> int:
        if self._comment != -1 and self._comment < k:
            self._comment = self.text.find("//", k)
        return self._comment

    def next(self, i: int):
        text = self.text
        n = len(text)
        while True:
            while i < n and text[i].isspace():
                i += 1
            if text.startswith("//", i):
                j = text.find("\n", i)
                i = n if j < 0 else j + 1
                continue
            break
        if i >= n:
            return None, n
        if text.startswith("gate", i) and (i + 4 < n and text[i + 4].isspace()):
            j = text.find("}", i)
            if j < 0:
                raise QasmError("unterminated gate definition")
            return _strip_comments(text[i:j + 1]), j + 1
        k = i
        while True:  # the terminating ';' is the first one that is not inside a comment
            j = text.find(";", k)
            c = self._next_comment(k)
            if c >= 0 and (j < 0 or c < j):
                e = text.find("\n", c)
                if e < 0:
                    j = -1
                    break
                k = e + 1
                continue
            break
        if j < 0:
            rest = _strip_comments(text[i:]).strip()
            if rest:
                raise QasmError(f"missing ';' after: {rest[:40]!r}")
            return None, n
        return _strip_comments(text[i:j]).strip(), j + 1


def _scan_plain_gates(circ: "Circuit", text: str, data: bytes, pos: int, qregs) -> int:
    """Hands the run of plain gate statements that starts at ``pos`` to the native scanner
    (qp_qasm_scan), appends it to ``circ`` as arrays and returns where the scanner stopped."""
    import ctypes as C

    from . import _lib
    L = _lib.lib()
    names_blob = b"".join(name.encode() + b"\0" for name, _ in qregs)
    sizes = np.array([sz for _, sz in qregs], np.int32)
    base = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32) if len(qregs) else np.zeros(0, np.int32)
    n = len(data)
    while True:
        cap = max(1024, min(1 << 20, (n - pos) // 8 + 16))
        h = np.empty(cap, np.uint64)
        npos = np.empty((cap, 2), np.int32)
        ppos = np.empty((cap, 2), np.int32)
        qidx = np.empty((cap, 3), np.int32)
        cnt, stop = C.c_int64(0), C.c_int64(0)
        _lib.check(L.qp_qasm_scan(data, n, pos, len(qregs), names_blob, _lib.ptr(sizes) if len(qregs) else None,
                                  _lib.ptr(base) if len(qregs) else None, cap, _lib.ptr(h), _lib.ptr(npos),
                                  _lib.ptr(ppos), _lib.ptr(qidx), C.byref(cnt), C.byref(stop)))
        k = cnt.value
        if k:
            uh, first, inv = np.unique(h[:k], return_index=True, return_inverse=True)
            names = [text[npos[f, 0]:npos[f, 0] + npos[f, 1]] for f in first.tolist()]
            kinds = [OpType.of(nm) if nm in _PYTKET_NAMES else OpType.CustomGate for nm in names]
            circ.add_run(_GateRun(text, inv.astype(np.int32), names, kinds, ppos[:k].copy(), qidx[:k].copy()))
        pos = stop.value
        if k < cap:
            return pos


def circuit_from_qasm_str(text: str, maxwidth: int = 32, native: bool = True) -> Circuit:
    circ = Circuit()
    qreg_size = {}
    creg_size = {}

    def resolve(arg: str):
        m = _ARG_RE.match(arg.strip())
        if not m:
            raise QasmError(f"bad argument {arg!r}")
        name, idx = m.group(1), m.group(2)
        if name in qreg_size:
            if idx is None:
                return "q", [circ.qubit(name, i) for i in range(qreg_size[name])], True
            if int(idx) >= qreg_size[name]:
                raise QasmError(f"index out of range: {arg}")
            return "q", [circ.qubit(name, int(idx))], False
        if name in creg_size:
            if idx is None:
                return "c", [(name, i) for i in range(creg_size[name])], True
            return "c", [(name, int(idx))], False
        raise QasmError(f"unknown register in {arg!r}")

    def emit(name, params, args, optype=None):
        resolved = [resolve(a) for a in args]
        width = max([len(r[1]) for r in resolved if r[2]], default=1)
        for r in resolved:
            if r[2] and len(r[1]) != width:
                raise QasmError("register size mismatch in broadcast")
        for k in range(width):
            qs, bs = [], []
            for kind, items, whole in resolved:
                it = items[k] if whole else items[0]
                (qs if kind == "q" else bs).append(it)
            circ.add_command(name, qs, params, optype=optype, bits=bs)

    # Plain gate statements (nearly all of a circuit file) go through the native scanner in runs; every
    # other statement is handled here, one at a time, between two runs.
    use_native = native and text.isascii()
    data = text.encode() if use_native else b""
    pos = 0
    statements = _Statements(text)
    while True:
        if use_native:
            pos = _scan_plain_gates(circ, text, data, pos, circ._qregs)
        st, pos = statements.next(pos)
        if st is None:
            break
        if not st:
            continue
        head = st.split(None, 1)[0]
        if head == "OPENQASM" or head == "include":
            continue
        if head in ("qreg", "creg"):
            m = re.match(r"^(qreg|creg)\s+([A-Za-z_][A-Za-z0-9_]*)\s*\[\s*(\d+)\s*\]$", st)
            if not m:
                raise QasmError(f"bad register declaration {st!r}")
            if m.group(1) == "qreg":
                qreg_size[m.group(2)] = int(m.group(3))
                circ.add_q_register(m.group(2), int(m.group(3)))
            else:
                if int(m.group(3)) > maxwidth:
                    raise QasmError(f"classical register wider than maxwidth={maxwidth}")
                creg_size[m.group(2)] = int(m.group(3))
            continue
        if head in ("gate", "opaque"):
            continue  # applications of user gates become one CustomGate command each
        cond = False
        if st.startswith("if"):
            m = re.match(r"^if\s*\(\s*[A-Za-z_][A-Za-z0-9_]*\s*==\s*\d+\s*\)\s*(.*)$", st, re.S)
            if not m:
                raise QasmError(f"bad conditional {st!r}")
            st, cond = m.group(1).strip(), True
            head = st.split(None, 1)[0]
        if head == "measure":
            m = re.match(r"^measure\s+(.+?)\s*->\s*(.+)$", st, re.S)
            if not m:
                raise QasmError(f"bad measure {st!r}")
            emit("measure", "", [m.group(1), m.group(2)],
                 optype=OpType.Conditional if cond else OpType.Measure)
            continue
        if head == "reset":
            emit("reset", "", [st[5:].strip()], optype=OpType.Conditional if cond else OpType.Reset)
            continue
        if head == "barrier":
            args = [a for a in st[7:].split(",") if a.strip()]
            qs = []
            for a in args:
                qs.extend(resolve(a)[1])
            circ.add_command("barrier", qs, optype=OpType.Barrier)
            continue
        m = _GATE_RE.match(st)
        if not m:
            raise QasmError(f"cannot parse {st!r}")
        name, params, rest = m.group(1), m.group(3) or "", m.group(4)
        args = [a for a in rest.split(",") if a.strip()]
        if not args:
            raise QasmError(f"gate without arguments: {st!r}")
        optype = OpType.Conditional if cond else (OpType.of(name) if name in _PYTKET_NAMES else OpType.CustomGate)
        emit(name, params, args, optype=optype)
    return circ


def circuit_from_qasm(path: str, maxwidth: int = 32) -> Circuit:
    """Same call shape as ``pytket.qasm.circuit_from_qasm`` (main.py:80)."""
    with open(path, "r") as f:
        return circuit_from_qasm_str(f.read(), maxwidth=maxwidth)


# ----------------------------------------------------------------------------------------------
# Synthetic circuits
# ----------------------------------------------------------------------------------------------

def layered_gate_arrays(Q: int, T: int, p2: float = 0.5, seed: int = 0):
    """Gate table of a random layered circuit, one ASAP layer per timeslice.

    Mirrors the gate mix of the reference generator's ``random_circuit_2n`` recipe
    (Circuit_Generator.py:8-25: random two-qubit pairings plus single-qubit gates) without qiskit:
    per layer draw a permutation of the Q qubits with ``numpy.random.Generator(PCG64(seed))``, pair
    the first ``2*floor(p2*Q/2)`` entries into ``cx`` gates and give every other qubit one
    single-qubit gate from {x, h, rz}.  Every qubit is busy in every layer, so the ASAP layering
    of the result is exactly these layers.

    Returns ``(q0, q1, kind)`` int32 arrays of length ``T * (Q - npairs)``; ``q1 = -1`` for
    single-qubit gates and ``kind`` indexes ``("cx", "x", "h", "rz")``.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    npairs = int(p2 * Q / 2)
    per_layer = Q - npairs
    q0 = np.empty(T * per_layer, np.int32)
    q1 = np.full(T * per_layer, -1, np.int32)
    kind = np.zeros(T * per_layer, np.int32)
    for t in range(T):
        perm = rng.permutation(Q).astype(np.int32)
        base = t * per_layer
        q0[base:base + npairs] = perm[0:2 * npairs:2]
        q1[base:base + npairs] = perm[1:2 * npairs:2]
        q0[base + npairs:base + per_layer] = perm[2 * npairs:]
        kind[base + npairs:base + per_layer] = rng.integers(1, 4, size=Q - 2 * npairs)
    return q0, q1, kind


_KIND_NAMES = ("cx", "x", "h", "rz")


def layered_circuit(Q: int, T: int, p2: float = 0.5, seed: int = 0) -> Circuit:
    """``layered_gate_arrays`` as a Circuit over one register ``q[Q]``."""
    q0, q1, kind = layered_gate_arrays(Q, T, p2, seed)
    circ = Circuit()
    circ.add_q_register("q", Q)
    qs = [circ.qubit("q", i) for i in range(Q)]
    for a, b, k in zip(q0.tolist(), q1.tolist(), kind.tolist()):
        if b >= 0:
            circ.add_command("cx", [qs[a], qs[b]])
        else:
            circ.add_command(_KIND_NAMES[k], [qs[a]], "pi/4" if k == 3 else "")
    return circ


def circuit_to_qasm_str(circ: Circuit) -> str:
    """Serialise (registers + commands) back to OpenQASM 2.0; classical bits of measures are kept."""
    lines = ["OPENQASM 2.0;", 'include "qelib1.inc";']
    for name, size in circ._qregs:
        lines.append(f"qreg {name}[{size}];")
    cregs = {}
    for c in circ:
        for (r, i) in c.bits:
            cregs[r] = max(cregs.get(r, 0), i + 1)
    for r, n in cregs.items():
        lines.append(f"creg {r}[{n}];")
    for c in circ:
        args = ",".join(str(q) for q in c.qubits)
        if c.op.name == "measure":
            r, i = c.bits[0]
            lines.append(f"measure {args} -> {r}[{i}];")
        elif c.op.params:
            lines.append(f"{c.op.name}({c.op.params}) {args};")
        else:
            lines.append(f"{c.op.name} {args};")
    return "\n".join(lines) + "\n"
