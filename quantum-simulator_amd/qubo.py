"""The QUBO placer's model (Qubo_placer.py:40-133) as integer tables.

Variables, per timeslice t with active qubits q_0 < q_1 < ... (sorted by ``str(qubit)``,
Qubo_placer.py:48) and C cores of capacity ``cap``:
    x[t, j, i] = 1 iff active qubit i sits on core j          (Binary 'x_{t}_{j}-{i}')
    y[t, j, p]   slack bits of core j, p < cap                  (Binary 'y_{t}_{j}-{p}')
Objective (weights Qubo_placer.py:29-32):
    4 * sum_i (sum_j x - 1)^2 + 4 * sum_j (sum_i x - sum_p y)^2
  + 10 * sum_j [ sum_i deg_i x_ji - 2 sum_{(i,l) interacting} x_ji x_jl ]           per slice
  + 10 * sum_{q in slices t and t+1} sum_{j,l} hops[j,l] x[t,j,i_q] x[t+1,l,i'_q]   between slices
Every coefficient is an integer, so energies are exact in int32.

Written as  E(s) = const + sum_v a_v s_v + sum_{v<w} S_vw s_v s_w  with S symmetric, zero diagonal:
    same qubit, two cores          S = +8            a(x) = 10 * deg_i
    same core, two qubits          S = +8 (-20 more if they interact)
    same core, two slack bits      S = +8            a(y) = +4
    same core, qubit and slack     S = -8            const = 4 * (active qubits) per slice
    same qubit in t and t+1        S = 10 * hops[j, l]
Variable order inside a slice: v = j * (n_t + cap) + k, k < n_t for x, k >= n_t for y.  Each slice's
block is padded to a multiple of 128 variables so the MFMA contraction needs no edge handling.
"""
from __future__ import annotations

import math

import numpy as np

PAD = 128
A_ASSIGN, B_CAP, C_LAP, D_MOVE = 4, 4, 10, 10


class QuboModel:
    def __init__(self, ir, C: int, cap: int, hops: np.ndarray, qubit_names=None):
        self.ir, self.C, self.cap = ir, int(C), int(cap)
        H = np.asarray(hops)
        if not np.isfinite(H).all():
            raise ValueError("QUBO placer needs a connected core graph (finite hop counts)")
        self.H = H.astype(np.int64)
        names = qubit_names if qubit_names is not None else (
            [str(q) for q in ir.qubits] if ir.qubits is not None else [f"q[{i}]" for i in range(ir.Q)])
        self.T = ir.T
        self.act, self.pos, self.inter, self.deg = [], [], [], []
        for layer in ir.slices:
            qs = sorted({q for g in layer for q in ir.gate_q[g]}, key=lambda q: names[q])
            pos = {q: i for i, q in enumerate(qs)}
            es = set()
            for g in layer:
                gq = ir.gate_q[g]
                for x in range(len(gq)):
                    for y in range(x + 1, len(gq)):
                        a, b = pos[gq[x]], pos[gq[y]]
                        if a != b:
                            es.add((min(a, b), max(a, b)))
            deg = np.zeros(len(qs), np.int32)
            for a, b in es:
                deg[a] += 1
                deg[b] += 1
            self.act.append(qs)
            self.pos.append(pos)
            self.inter.append(sorted(es))
            self.deg.append(deg)
        self.n = np.array([len(a) for a in self.act], np.int32)
        self.W = self.n + self.cap                      # variables per core in slice t
        self.Nt = self.C * self.W                       # variables in slice t
        self.Np = (self.Nt + PAD - 1) // PAD * PAD      # padded
        self.off = np.concatenate([[0], np.cumsum(self.Np)]).astype(np.int64)
        self.N = int(self.Nt.sum())
        self.Npad = int(self.off[-1])
        self.const = int(A_ASSIGN * self.n.sum())
        # same qubit in the next slice: nxt[t][i] = local index in slice t+1 or -1
        self.nxt, self.prv = [], []
        for t in range(self.T):
            nx = np.full(self.n[t], -1, np.int32)
            pv = np.full(self.n[t], -1, np.int32)
            if t + 1 < self.T:
                for i, q in enumerate(self.act[t]):
                    nx[i] = self.pos[t + 1].get(q, -1)
            if t > 0:
                for i, q in enumerate(self.act[t]):
                    pv[i] = self.pos[t - 1].get(q, -1)
            self.nxt.append(nx)
            self.prv.append(pv)

    # -- index helpers ----------------------------------------------------------------------------
    def var_x(self, t, j, i):
        return int(self.off[t] + j * self.W[t] + i)

    def var_y(self, t, j, p):
        return int(self.off[t] + j * self.W[t] + self.n[t] + p)

    # -- dense tables for the contraction -----------------------------------------------------------
    def linear(self) -> np.ndarray:
        """a_v over the padded variable axis (int32)."""
        a = np.zeros(self.Npad, np.int32)
        for t in range(self.T):
            n, W = int(self.n[t]), int(self.W[t])
            blk = np.zeros((self.C, W), np.int32)
            blk[:, :n] = C_LAP * self.deg[t][None, :]
            blk[:, n:] = B_CAP
            a[self.off[t]:self.off[t] + self.Nt[t]] = blk.reshape(-1)
        return a

    def slice_block(self, t) -> np.ndarray:
        """S restricted to slice t: int8 [Np_t, Np_t], symmetric, zero diagonal."""
        n, W, C = int(self.n[t]), int(self.W[t]), self.C
        same = np.full((W, W), 2 * B_CAP, np.int32)
        same[:n, n:] = -2 * B_CAP
        same[n:, :n] = -2 * B_CAP
        for a, b in self.inter[t]:
            same[a, b] -= 2 * C_LAP
            same[b, a] -= 2 * C_LAP
        np.fill_diagonal(same, 0)
        cross = np.zeros((W, W), np.int32)
        cross[np.arange(n), np.arange(n)] = 2 * A_ASSIGN
        eye = np.eye(C, dtype=np.int32)
        S = np.kron(eye, same) + np.kron(1 - eye, cross)
        out = np.zeros((int(self.Np[t]), int(self.Np[t])), np.int8)
        assert np.abs(S).max() <= 127
        out[:S.shape[0], :S.shape[1]] = S
        return out

    def move_layers(self, t):
        """The coupling between slice t (rows) and t+1 (columns), 10 * hops on matching qubits, as a
        list of int8 [Np_t, Np_{t+1}] layers whose sum is the block (one layer when 10*hops <= 127)."""
        n0, n1, W0, W1, C = int(self.n[t]), int(self.n[t + 1]), int(self.W[t]), int(self.W[t + 1]), self.C
        P = np.zeros((W0, W1), np.int32)
        for i in range(n0):
            k = self.nxt[t][i]
            if k >= 0:
                P[i, k] = 1
        full = np.kron(D_MOVE * self.H.astype(np.int32), P)
        layers = []
        rest = full
        while True:
            part = np.minimum(rest, 127)
            out = np.zeros((int(self.Np[t]), int(self.Np[t + 1])), np.int8)
            out[:full.shape[0], :full.shape[1]] = part
            layers.append(out)
            rest = rest - part
            if not rest.any():
                break
        return layers

    # -- states -----------------------------------------------------------------------------------
    def state_from_rows(self, rows, slack="fill") -> np.ndarray:
        """int8 [Npad] state that puts active qubit i of slice t on core rows[t][q] (one-hot) and
        sets as many slack bits of each core as it holds qubits (capacity term zero when it fits)."""
        s = np.zeros(self.Npad, np.int8)
        for t in range(self.T):
            load = np.zeros(self.C, np.int64)
            for i, q in enumerate(self.act[t]):
                c = rows[t][q]
                if c is None or c == 0xFF:
                    continue
                s[self.var_x(t, int(c), i)] = 1
                load[int(c)] += 1
            if slack == "fill":
                for j in range(self.C):
                    for p in range(min(self.cap, int(load[j]))):
                        s[self.var_y(t, j, p)] = 1
        return s

    def decode(self, state: np.ndarray):
        """Qubo_placer.py:146-172: per slice and active qubit (in sorted order) the first core whose
        bit is set wins unless that core is already full; the rest go, in the same order, to the
        least-loaded core (first minimum).  Returns T rows of length Q with None for inactive qubits."""
        rows = []
        for t in range(self.T):
            n, W = int(self.n[t]), int(self.W[t])
            blk = state[self.off[t]:self.off[t] + self.Nt[t]].reshape(self.C, W)[:, :n]
            row = [None] * self.ir.Q
            load = [0] * self.C
            pending = []
            for i, q in enumerate(self.act[t]):
                on = np.flatnonzero(blk[:, i] == 1)
                j = int(on[0]) if on.size else None
                if j is not None and load[j] < self.cap:
                    row[q] = j
                    load[j] += 1
                else:
                    pending.append(q)
            for q in pending:
                j = min(range(self.C), key=lambda jj: load[jj])
                row[q] = j
                load[j] += 1
            rows.append(row)
        return rows

    # -- annealing schedule -----------------------------------------------------------------------
    def beta_range(self):
        """(hot, cold) inverse temperatures in the spirit of the D-Wave neal default: hot enough that
        the largest single-flip energy change is accepted half the time, cold enough that the smallest
        one (4, the constraint quantum) is accepted 1% of the time."""
        worst = 0
        for t in range(self.T):
            n = int(self.n[t])
            dmax = int(self.deg[t].max()) if n else 0
            move = 2 * D_MOVE * int(self.H.max())
            worst = max(worst, 2 * A_ASSIGN * self.C + 2 * B_CAP * (n + self.cap) + C_LAP * 3 * dmax + move)
        return math.log(2.0) / max(worst, 1), math.log(100.0) / 4.0


class DeviceQubo:
    """A QuboModel uploaded to one GPU, plus the three device entry points of the QUBO path."""

    def __init__(self, model: QuboModel, device=None, dense=True):
        import ctypes as C
        import torch
        from . import _lib
        from .engine import default_device
        self._C, self._lib, self._torch = C, _lib, torch
        self.model = m = model
        self.device = device if device is not None else default_device()
        dev = self.device
        self._keep = []

        def up(arr):
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
            self._keep.append(t)
            return t

        off = up(m.off.astype(np.int64))
        self.dense = None
        if dense:
            segs, seg_ptr = [], [0]
            move = [m.move_layers(t) for t in range(m.T - 1)]
            for t in range(m.T):
                blk = up(m.slice_block(t))
                segs.append((blk.data_ptr(), int(m.Np[t]), int(m.Np[t]), int(m.off[t])))
                if t + 1 < m.T:
                    for layer in move[t]:
                        d = up(layer)
                        segs.append((d.data_ptr(), int(m.Np[t + 1]), int(m.Np[t + 1]), int(m.off[t + 1])))
                if t > 0:
                    for layer in move[t - 1]:
                        d = up(np.ascontiguousarray(layer.T))
                        segs.append((d.data_ptr(), int(m.Np[t - 1]), int(m.Np[t - 1]), int(m.off[t - 1])))
                seg_ptr.append(len(segs))
            seg_arr = (_lib.QpQuboSeg * len(segs))(*[_lib.QpQuboSeg(*s) for s in segs])
            seg_dev = up(np.frombuffer(bytes(seg_arr), np.uint8).copy())
            tile_slice = np.repeat(np.arange(m.T, dtype=np.int32), (m.Np // PAD).astype(np.int64))
            self.dense = _lib.QpQuboDense(m.T, m.Npad, m.const, 0, seg_dev.data_ptr(),
                                          up(np.asarray(seg_ptr, np.int32)).data_ptr(), up(tile_slice).data_ptr(),
                                          off.data_ptr(), up(m.linear()).data_ptr())
            self.dense_bytes = sum(int(t.numel() * t.element_size()) for t in self._keep)
        nmax = int(max(1, m.n.max()))
        deg = np.zeros((m.T, nmax), np.uint8)
        nxt = np.full((m.T, nmax), -1, np.int16)
        prv = np.full((m.T, nmax), -1, np.int16)
        ptr = np.zeros(m.T * nmax + 1, np.int32)
        nbr = []
        for t in range(m.T):
            n = int(m.n[t])
            deg[t, :n] = m.deg[t]
            nxt[t, :n] = m.nxt[t]
            prv[t, :n] = m.prv[t]
            partners = [[] for _ in range(n)]
            for a, b in m.inter[t]:
                partners[a].append(b)
                partners[b].append(a)
            for i in range(nmax):
                if i < n:
                    nbr.extend(partners[i])
                ptr[t * nmax + i + 1] = len(nbr)
        if m.H.max() > 255:
            raise ValueError("hop counts above 255 are not supported")
        # word indices into the [T][nmax] table; T * nmax = "none"
        none = m.T * nmax
        if none > 65534:
            raise ValueError("T * max active qubits must stay below 65535")
        first = np.full((m.T, nmax), none, np.uint64)
        for t in range(m.T):
            for i in range(int(m.n[t])):
                a, b = ptr[t * nmax + i], ptr[t * nmax + i + 1]
                if b > a:
                    first[t, i] = t * nmax + nbr[a]
        trow = (np.arange(m.T, dtype=np.int64) * nmax)[:, None]
        nxt_w = np.where(nxt >= 0, trow + nmax + nxt, none).astype(np.uint64)
        prv_w = np.where(prv >= 0, trow - nmax + prv, none).astype(np.uint64)
        info = (deg.astype(np.uint64) | (first << np.uint64(16)) | (nxt_w << np.uint64(32)) | (prv_w << np.uint64(48)))
        self.struct = _lib.QpQuboStruct(m.T, m.C, m.cap, nmax, m.Npad, up(m.n.astype(np.int32)).data_ptr(),
                                        off.data_ptr(), up(info).data_ptr(), up(ptr).data_ptr(),
                                        up(np.asarray(nbr if nbr else [0], np.int16)).data_ptr(),
                                        up(m.H.astype(np.uint8)).data_ptr())

    def _stream(self):
        return self._C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _check_states(self, Xt):
        torch = self._torch
        if Xt.dtype != torch.int8 or Xt.dim() != 2 or Xt.shape[1] != self.model.Npad or not Xt.is_contiguous():
            raise ValueError("states must be a contiguous int8 [R, Npad] tensor")
        return Xt.shape[0]

    def energy(self, Xt):
        """Exact energies (int32 [R]) of replica states Xt (int8 [R, Npad], R % 128 == 0)."""
        R = self._check_states(Xt)
        out = self._torch.empty(R, dtype=self._torch.int32, device=self.device)
        with self._torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().qp_qubo_energy_i32(self._C.byref(self.dense), Xt.data_ptr(), R,
                                                               out.data_ptr(), self._stream()))
        return out

    def fields(self, Xt):
        """Local fields F[v, r] (int32 [Npad, R])."""
        R = self._check_states(Xt)
        out = self._torch.empty((self.model.Npad, R), dtype=self._torch.int32, device=self.device)
        with self._torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().qp_qubo_fields_i32(self._C.byref(self.dense), Xt.data_ptr(), R,
                                                               out.data_ptr(), self._stream()))
        return out

    def anneal(self, R, sweeps, seed=42, beta=None, init=None):
        """R replicas of simulated annealing; returns (states int8 [R, Npad], summed dE int32 [R])."""
        torch = self._torch
        hot, cold = beta if beta is not None else self.model.beta_range()
        flags = 0
        if init is not None:
            Xt = init.clone()
            flags = self._lib.QP_QUBO_INIT_GIVEN
        else:
            Xt = torch.zeros((R, self.model.Npad), dtype=torch.int8, device=self.device)
        dE = torch.empty(R, dtype=torch.int32, device=self.device)
        with self._torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().qp_qubo_anneal(self._C.byref(self.struct), R, sweeps, hot, cold, seed,
                                                           Xt.data_ptr(), dE.data_ptr(), flags, self._stream()))
        return Xt, dE
