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
    same core, two qubits          S = +8 (-20 more if they interact)      a(x) = 10 * deg_i
    same core, two slack bits      S = +8                                  a(y) = +4
    same core, qubit and slack     S = -8                                  const = 4 * (active qubits) per slice
    same qubit, two cores          S = +8
    same qubit in t and t+1        S = 10 * hops[j, l]
The slice's matrix is a Kronecker sum: the ``same core`` couplings are one W x W matrix (W = n_t + cap)
repeated for every core, the ``same qubit`` couplings a diagonal repeated for every pair of cores, and the
coupling between slices a permutation of matching qubits times the C x C hop matrix.  Only the W x W
matrix goes through the MFMA contraction; the other two are evaluated from per-qubit core bit masks.

Variable order inside a slice: v = off[t] + j * P_t + k, k < n_t for x, n_t <= k < W for y, with the
per-core pitch P_t = W rounded up to 32 (the rest is padding that stays zero); each slice's block is padded
to a multiple of 128 variables.
"""
from __future__ import annotations

import math

import numpy as np

PAD = 128
PITCH = 32
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
        self.P = (self.W + PITCH - 1) // PITCH * PITCH  # their pitch in the state vector
        self.Nt = self.C * self.P                       # state bytes of slice t (padding inside included)
        self.Np = (self.Nt + PAD - 1) // PAD * PAD      # padded
        self.off = np.concatenate([[0], np.cumsum(self.Np)]).astype(np.int64)
        self.N = int((self.C * self.W).sum())           # binary variables of the model
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
        return int(self.off[t] + j * self.P[t] + i)

    def var_y(self, t, j, p):
        return int(self.off[t] + j * self.P[t] + self.n[t] + p)

    # -- tables for the contraction ------------------------------------------------------------------
    def real_mask(self) -> np.ndarray:
        """bool [Npad]: which positions of the padded state vector are variables of the model."""
        m = np.zeros(self.Npad, bool)
        for t in range(self.T):
            blk = np.zeros((self.C, int(self.P[t])), bool)
            blk[:, :int(self.W[t])] = True
            m[self.off[t]:self.off[t] + self.Nt[t]] = blk.reshape(-1)
        return m

    def linear(self) -> np.ndarray:
        """a_v over the padded variable axis (int32)."""
        a = np.zeros(self.Npad, np.int32)
        for t in range(self.T):
            n, W, P = int(self.n[t]), int(self.W[t]), int(self.P[t])
            blk = np.zeros((self.C, P), np.int32)
            blk[:, :n] = C_LAP * self.deg[t][None, :]
            blk[:, n:W] = B_CAP
            a[self.off[t]:self.off[t] + self.Nt[t]] = blk.reshape(-1)
        return a

    def same_core_block(self, t) -> np.ndarray:
        """The couplings between the W variables of one core in slice t (the same for every core): int8,
        zero diagonal, padded with zeros to [rows: P_t up to 128][columns: P_t up to 128]."""
        n, W, P = int(self.n[t]), int(self.W[t]), int(self.P[t])
        same = np.full((W, W), 2 * B_CAP, np.int32)
        same[:n, n:] = -2 * B_CAP
        same[n:, :n] = -2 * B_CAP
        for a, b in self.inter[t]:
            same[a, b] -= 2 * C_LAP
            same[b, a] -= 2 * C_LAP
        np.fill_diagonal(same, 0)
        assert np.abs(same).max() <= 127
        Pp = (P + PAD - 1) // PAD * PAD
        out = np.zeros((Pp, Pp), np.int8)
        out[:W, :W] = same
        return out

    def coupling_counts(self):
        """(non-zero couplings S_vw, ordered pairs; multiply-accumulates per replica the MFMA contraction runs;
        what a dense contraction of the whole block-tridiagonal matrix would run)."""
        nnz = mfma = dense = 0
        for t in range(self.T):
            n, W, P, C = int(self.n[t]), int(self.W[t]), int(self.P[t]), self.C
            nnz += C * (W * W - W) + n * C * (C - 1)
            Pp = (P + PAD - 1) // PAD * PAD
            mfma += C * Pp * P
            width = int(self.Np[t]) + (int(self.Np[t + 1]) if t + 1 < self.T else 0) + (int(self.Np[t - 1]) if t > 0 else 0)
            dense += int(self.Np[t]) * width
            if t + 1 < self.T:
                nnz += 2 * int((self.nxt[t] >= 0).sum()) * int((self.H != 0).sum())
        return nnz, mfma, dense

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
            n, P = int(self.n[t]), int(self.P[t])
            blk = state[self.off[t]:self.off[t] + self.Nt[t]].reshape(self.C, P)[:, :n]
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

    def __init__(self, model: QuboModel, device=None):
        import ctypes as C
        import torch
        from . import _lib
        from .engine import default_device
        self._C, self._lib, self._torch = C, _lib, torch
        self.model = m = model
        self.device = device if device is not None else default_device()
        dev = self.device
        self._keep = []
        self._ws = None

        def up(arr):
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
            self._keep.append(t)
            return t

        if m.C > 32 or m.cap > 32:
            raise ValueError("the QUBO kernels keep a core mask per qubit and a slack mask per core: at most 32 cores, capacity 32")
        off = up(m.off.astype(np.int64))
        nmax = int(max(1, m.n.max()))
        blocks = [m.same_core_block(t) for t in range(m.T)]
        same_off = np.concatenate([[0], np.cumsum([b.size for b in blocks])]).astype(np.int64)
        same = up(np.concatenate([b.reshape(-1) for b in blocks]) if blocks else np.zeros(1, np.int8))
        row_tiles = np.array([b.shape[0] // PAD for b in blocks], np.int32)        # 128-row tiles of the W x W block
        tile_first = np.concatenate([[0], np.cumsum(m.C * row_tiles)]).astype(np.int32)
        tile_slice = np.repeat(np.arange(m.T, dtype=np.int32), (m.C * row_tiles).astype(np.int64))
        nxt = np.full((m.T, nmax), -1, np.int32)
        prv = np.full((m.T, nmax), -1, np.int32)
        for t in range(m.T):
            nxt[t, :int(m.n[t])] = m.nxt[t]
            prv[t, :int(m.n[t])] = m.prv[t]
        self.kron = _lib.QpQuboKron(
            m.T, m.C, m.cap, nmax, m.Npad, m.const, int(tile_first[-1]),
            up(m.n.astype(np.int32)).data_ptr(), up(m.P.astype(np.int32)).data_ptr(), off.data_ptr(),
            same.data_ptr(), up(same_off).data_ptr(), up(tile_slice if len(tile_slice) else np.zeros(1, np.int32)).data_ptr(),
            up(tile_first).data_ptr(), up(m.linear()).data_ptr(), up(nxt).data_ptr(), up(prv).data_ptr(),
            up((D_MOVE * m.H).astype(np.int32)).data_ptr())
        self.model_bytes = sum(int(t.numel() * t.element_size()) for t in self._keep)
        nxt = np.full((m.T, nmax), -1, np.int16)
        prv = np.full((m.T, nmax), -1, np.int16)
        deg = np.zeros((m.T, nmax), np.uint8)
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
                                        up(m.H.astype(np.uint8)).data_ptr(), up(m.P.astype(np.int32)).data_ptr())

    def _stream(self):
        return self._C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _check_states(self, Xt):
        torch = self._torch
        if Xt.dtype != torch.int8 or Xt.dim() != 2 or Xt.shape[1] != self.model.Npad or not Xt.is_contiguous():
            raise ValueError("states must be a contiguous int8 [R, Npad] tensor")
        return Xt.shape[0]

    def _workspace(self, R):
        need = int(self._lib.lib().qp_qubo_workspace(self._C.byref(self.kron), R))
        if self._ws is None or self._ws.numel() < need:
            self._ws = self._torch.empty(need, dtype=self._torch.uint8, device=self.device)
        return self._ws

    def energy(self, Xt):
        """Exact energies (int32 [R]) of replica states Xt (int8 [R, Npad], R % 128 == 0)."""
        R = self._check_states(Xt)
        out = self._torch.empty(R, dtype=self._torch.int32, device=self.device)
        ws = self._workspace(R)
        with self._torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().qp_qubo_energy_i32(self._C.byref(self.kron), Xt.data_ptr(), R,
                                                               out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return out

    def fields(self, Xt):
        """Local fields F[v, r] (int32 [Npad, R])."""
        R = self._check_states(Xt)
        out = self._torch.empty((self.model.Npad, R), dtype=self._torch.int32, device=self.device)
        ws = self._workspace(R)
        with self._torch.cuda.device(self.device):
            self._lib.check(self._lib.lib().qp_qubo_fields_i32(self._C.byref(self.kron), Xt.data_ptr(), R,
                                                               out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
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
