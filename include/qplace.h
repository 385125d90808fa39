/*
 * qplace.h — C ABI of libqplace.so: the MI355X (gfx950) placement-cost engine.
 *
 * The reference (surankan-de/quantum-simulator) is pure Python and has no FFI; its plug-in seam is
 * the pair of methods every placer exposes (main.py:85-97):
 *     place_per_timeslice(circuit, multicore_arch) -> list[dict[qubit -> core]]
 *     per_timeslice_cost(circuit, partition, multicore_arch) -> float
 * Each entry point below replaces one piece of what those methods compute; the comment on each names
 * the reference lines it stands in for; quantum-simulator_amd/_lib.py is the complete ctypes binding.
 *
 * Conventions
 *  - Plain pointers and sizes only.  "dev" pointers are HIP device memory owned by the caller;
 *    "host" pointers are ordinary memory.  The library never allocates device memory.
 *  - Every function returns QP_OK (0) or a negative QP_E* code; qp_last_error() then holds a message
 *    (thread-local).  Nothing throws or exits across the ABI.
 *  - Device entry points take a hipStream_t as void* and are asynchronous with respect to the host;
 *    the caller selects the device (hipSetDevice) before the call.
 *  - Core ids are uint8: up to 256 cores for dense partitions; sparse partitions reserve 0xFF for
 *    "qubit not in this slice's dict" and so hold up to 255.
 */
#ifndef QPLACE_H
#define QPLACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QP_VERSION 204

#define QP_OK 0
#define QP_EINVAL (-1)   /* bad argument (shape, alignment, range) */
#define QP_EHIP (-2)     /* a HIP runtime call failed */
#define QP_ENOSPC (-3)   /* caller-provided buffer or workspace too small */
#define QP_EINFEASIBLE (-4) /* assignment problem has no finite solution */
#define QP_ENOMEM (-5)   /* a host allocation (or a host worker thread's) failed */

#define QP_ABSENT 0xFF

/* One counted two-qubit gate (not Measure/Reset) of timeslice t, in the reference's slice order
 * (Placer.py:14-61 decides membership and order; HQA_placer.py:66-67 decides "counted"). */
typedef struct QpEdge {
    uint16_t u, v; /* qubit indices = positions in circuit.qubits (HQA_placer.py:32) */
    int32_t t;     /* timeslice */
} QpEdge;

/* One (timeslice, gate, qubit) visit of the gate-ordered walk the BFS/Random/QUBO/Base cost functions
 * do (BFS_placer.py:214-237).  prev_t is the previous timeslice in which the same qubit was visited
 * (-1: none): "last" in those loops is the assignment at that slice. */
typedef struct QpVisit {
    uint16_t q;       /* this qubit */
    uint16_t partner; /* the gate's other qubit if this is the first qubit of a counted two-qubit gate,
                         else 0xFFFF */
    int32_t t;
    int32_t prev_t;
    int32_t pad_;
} QpVisit;

/* Device-resident circuit tables (all pointers dev, built once per circuit by the host side). */
typedef struct QpCircuit {
    int32_t Q;  /* qubits */
    int32_t T;  /* timeslices */
    int32_t nE; /* counted two-qubit gates */
    int32_t nV; /* visits */
    int32_t max_slice_edges; /* most counted gates in one slice; 0 = not given (Q/2 is assumed) */
    int32_t group_ldq;       /* row pitch group_sched was laid out for; 0 = no grouped gate list */
    const int32_t* slice_edge_ptr;  /* [T+1] CSR over edges */
    const QpEdge* edges;            /* [nE] */
    const int16_t* lookahead;       /* [nE*6] other-qubit of the <=2 two-qubit commands that touch
                                       (u,v) in slices t+1, t+2, t+3 (slots 2k, 2k+1 = slice t+1+k, in
                                       slice order), -1 = empty (HQA_placer.py:227-237) */
    const int32_t* slice_visit_ptr; /* [T+1] CSR over visits */
    const QpVisit* visits;          /* [nV] */
    const uint32_t* edges_sched;    /* optional: u | v << 16 of the same gates in the layout qp_schedule_gates
                                       writes (every slice padded with zero words to a multiple of
                                       QP_SCHED_CHUNK, QP_SCHED_TAIL zero words after the last slice); the dense
                                       cost kernels for long rows read this list */
    const int32_t* sched_ptr;       /* [T+1] start of every slice in edges_sched (required with it) */
    const uint32_t* group_sched;    /* optional, rows of at most 512 bytes: the same gates in the layout
                                       qp_schedule_gate_groups writes for the row pitch group_ldq (the gates of
                                       qp_group_rows(group_ldq) consecutive slices together, endpoints as byte offsets
                                       into those rows, padded like edges_sched); the dense cost kernels for short
                                       rows read this list when the partitions' row pitch equals group_ldq */
    const int32_t* group_ptr;       /* [ceil(T / qp_group_rows(group_ldq)) + 1] start of every group (required with it) */
    const uint8_t* visit_mask;      /* optional, rows of up to 1024 bytes: [T][visit_mask_ldq] bytes, 0xFF where the
                                       qubit is an operand of some command of the slice (a visit of the gate-ordered
                                       walk), 0 elsewhere and in the padding; with it and edges_sched, large batches
                                       of the gate-ordered cost are walked row by row instead of visit by visit;
                                       with edges and slice_edge_ptr likewise for rows of at most 512 bytes (which
                                       need visit_prev too); 16-byte aligned */
    int32_t visit_mask_ldq;         /* row pitch visit_mask was laid out for; 0 = none */
    int32_t reserved_;
    const int16_t* visit_prev;      /* optional with visit_mask, T <= 32767: [T][visit_mask_ldq], the last slice before
                                       slice t that visits the qubit, -1 if none; lets the walk start in the middle of
                                       a partition (small batches split into chunks of slices, fp64 tables counted
                                       by core pair); 32-byte aligned */
} QpCircuit;

/* Distance table as the kernels consume it. */
#define QP_DT_F64 0 /* C*C doubles, row = source core */
#define QP_DT_U16 1 /* C*C uint16: every finite entry is an integer <= 65534 (integer edge weights) */
/* Closed form of an integer table on a grid of cores (mesh, line, torus with equal integer edge weights):
 *   x = core & ((1 << xbits) - 1), y = core >> xbits,
 *   D[a][b] = wx * dx + wy * dy,  dx = |xa - xb|, or min(|xa - xb|, wrap_x - |xa - xb|) when wrap_x != 0
 * (dy likewise with wrap_y).  Filled by qp_dtable_grid_metric, which accepts a matrix only if every
 * entry equals the formula; with valid != 0 the dense cost kernel computes distances from core ids and
 * never reads the table.  No reference counterpart (it always queries the graph, Placer.py:63-76). */
typedef struct QpGridMetric {
    int32_t valid;
    int32_t xbits;
    uint32_t wx, wy;
    int32_t wrap_x, wrap_y; /* 0, or the ring length of that dimension (a power of two, 4..128) */
} QpGridMetric;

typedef struct QpDTable {
    int32_t kind;
    int32_t C;
    const void* table; /* dev, exactly C*C entries, row-major, zero diagonal; 16-byte aligned */
    QpGridMetric grid; /* all zero when unused; only consulted for QP_DT_U16 */
    int32_t symmetric; /* QP_DT_F64 only: non-zero = qp_dtable_symmetric accepted the matrix.  The dense cost
                          kernel may then use (D[a][b] + D[b][a]) / 2 for both directions of a pair, which is
                          what lets a table of more than 134 cores work out of on-chip memory (pair counts
                          instead of look-ups).  0 is always correct. */
    int32_t reserved_;
    double grid_scale; /* QP_DT_F64 only: when positive and grid.valid is set, every entry of the table equals
                          grid_scale * (grid.wx * dx + grid.wy * dy) to within a few units in the last place
                          (qp_dtable_scaled_grid_metric checked each one): a network whose links all carry the same
                          real weight, such as the two-core mesh of example_config.txt:1-12 with its one link
                          (multicore.py:159-161 draws it from random.uniform).  The dense cost kernel may then count
                          hops with the closed form and multiply once per partition.  0 is always correct. */
} QpDTable;

int qp_version(void);
const char* qp_last_error(void);

/* ---------------------------------------------------------------- host-side pieces */

/* Host threads the calling process should use for one call: min(limit, budget), at least 1.  The budget is the
 * number of CPUs the process may run on (its affinity mask) divided by the number of ranks that share the
 * node (LOCAL_WORLD_SIZE in the environment, as torch.distributed.run exports it; 1 when absent), unless
 * qp_set_host_threads gave a positive number (0 restores the default).  Every call below that chooses its own
 * thread count (qp_place_sparse) goes through it, and the Python side asks it for the default of every
 * `threads` / `lanes` argument, so that N ranks on one node do not start N times the node's cores in threads.
 * No reference counterpart: the reference is one Python thread (main.py:85-105 runs its repeats one by one). */
int32_t qp_host_threads(int32_t limit);
int qp_set_host_threads(int32_t n);

/* All-pairs core distances, host.  Row i is a Dijkstra from source i that accumulates dist[u] + w
 * outwards; unreachable pairs fall back to hop counts, then +inf; the diagonal is 0.
 * Replaces BasePlacer.get_weighted_distance (Placer.py:63-76) and the C*C fills built on it
 * (HQA_placer.py:47-54, TRHQA_placer.py:51-59, GTRHQA_placer.py:51-58). */
int qp_apsp_f64(int32_t C, int32_t n_edges, const int32_t* eu, const int32_t* ev, const double* w,
                double* D_out /* host [C*C] */);

/* The same for S networks that share their links (eu, ev) and differ in the weights: w [S][n_edges],
 * D_out [S][C*C], on `threads` host threads.  This is the distance work of simulate_broken_links
 * (HQA_placer.py:325-361 and its copies), which re-places the circuit on many copies of the network with some
 * link weights set to infinity: an infinite weight keeps the link traversable at infinite cost, as networkx
 * treats it. */
int qp_apsp_f64_batch(int32_t C, int32_t n_edges, const int32_t* eu, const int32_t* ev, const double* w, int32_t S,
                      double* D_out /* host [S][C*C] */, int32_t threads);

/* Unweighted hop counts as doubles (+inf when unreachable), host.  Replaces the
 * nx.shortest_path_length calls without weight (Qubo_placer.py:35-38,192,201). */
int qp_hops_f64(int32_t C, int32_t n_edges, const int32_t* eu, const int32_t* ev,
                double* H_out /* host [C*C] */);

/* Rectangular linear sum assignment, host; same result (including ties) as
 * scipy.optimize.linear_sum_assignment (HQA_placer.py:126, TRHQA_placer.py:171).
 * row_out/col_out receive min(nr,nc) pairs, rows ascending. */
int qp_lsap_f64(int32_t nr, int32_t nc, const double* cost /* host [nr*nc] */, int32_t* row_out,
                int32_t* col_out);

/* n independent problems solved on `threads` host threads.  Problem k reads cost + k*stride (row pitch
 * ld) with shape nr[k] x nc[k]; writes row_out/col_out + k*out_stride; status[k] is QP_OK or
 * QP_EINFEASIBLE. */
int qp_lsap_f64_batch(int32_t n, const int32_t* nr, const int32_t* nc, const double* cost, int64_t stride,
                      int32_t ld, int32_t* row_out, int32_t* col_out, int32_t out_stride, int32_t* status,
                      int32_t threads);

/* ASAP timeslices of a circuit of one- and two-qubit gates (gq1[g] = -1 for one qubit), host.
 * Same slice membership AND same order inside every slice as BasePlacer.break_into_timeslices
 * (Placer.py:14-61), whose order depends on how CPython iterates its successor sets; that iteration
 * order is reproduced.  order_out[n_gates]: gate indices slice after slice; slice_ptr_out[<= n_gates+1];
 * *T_out: number of slices. */
int qp_timeslices(int32_t n_gates, const int32_t* gq0, const int32_t* gq1, int32_t Q, int32_t* order_out,
                  int32_t* slice_ptr_out, int32_t* T_out);

/* OpenQASM 2.0 reader, the fast part: scans consecutive plain gate statements
 *     name [ ( parameters ) ] reg[i] { , reg[j] } ;        (one to three arguments, each one qubit)
 * from text[start..n), skipping white space and // comments, and stops in front of the first statement
 * that is anything else (a declaration, include, gate definition, measure, reset, barrier, if, an
 * argument naming a whole register or an unknown one, an index out of range): *stop_out is that
 * statement's offset (n at the end of the text) and the caller handles it before scanning on.
 * Quantum registers known so far: n_regs NUL-terminated names back to back in reg_names, their sizes,
 * and the global index of their first qubit.  Per gate k < *n_out: name_hash[k] (FNV-1a of the name),
 * name_pos[2k..] = offset, length of the name; param_pos[2k..] = offset, length of the text between the
 * parentheses (-1, -1 without parameters); qidx[3k..] = global qubit indices, -1 for unused slots.
 * Stands in for the per-statement work of pytket.qasm.circuit_from_qasm as main.py:80 uses it. */
int qp_qasm_scan(const char* text, int64_t n, int64_t start, int32_t n_regs, const char* reg_names,
                 const int32_t* reg_size, const int32_t* reg_base, int64_t cap, uint64_t* name_hash,
                 int32_t* name_pos, int32_t* param_pos, int32_t* qidx, int64_t* n_out, int64_t* stop_out);

/* Lays the counted gates of every slice out for QpCircuit.edges_sched / sched_ptr: words u | v << 16, each
 * slice padded with zero words (qubit 0 with itself: distance zero) to a multiple of QP_SCHED_CHUNK so that
 * a wave reads whole chunks without masking, QP_SCHED_TAIL zero words after the last slice so that reading
 * a few chunks ahead stays inside the buffer.  Inside a slice, runs of 32 gates get endpoints in distinct
 * 4-byte LDS banks where possible.  sched_capacity (words) must be at least
 * sum_t roundup(E_t, QP_SCHED_CHUNK) + QP_SCHED_TAIL.  No reference counterpart (the cost is a sum over a
 * slice's gates; their order is free). */
#define QP_SCHED_CHUNK 64
#define QP_SCHED_TAIL 512
int qp_schedule_gates(int32_t T, const int32_t* slice_edge_ptr, const uint16_t* eu, const uint16_t* ev,
                      int32_t* sched_ptr_out /* [T+1] */, uint32_t* sched_out, int64_t sched_capacity);

/* Rows of at most 512 bytes: a wave of the dense cost kernels takes G = qp_group_rows(ldq) = 64 / (ldq / 16)
 * consecutive slices per step (64 lanes x 16 bytes of rows), slices j*G .. j*G + G - 1 forming group j.
 * qp_schedule_gate_groups lays the counted gates out group by group for QpCircuit.group_sched / group_ptr: words
 * a | b << 16 with a = (t - j*G) * ldq + u and b = (t - j*G) * ldq + v, the byte offsets of the gate's two qubits
 * in the group's rows, every group padded with zero words to a multiple of QP_SCHED_CHUNK, QP_SCHED_TAIL zero
 * words at the end, runs of 32 gates spread over LDS banks as in qp_schedule_gates.  sched_capacity (words)
 * must be at least sum_j roundup(E_j, QP_SCHED_CHUNK) + QP_SCHED_TAIL.  qp_group_rows returns 0 for a row pitch
 * that is not a multiple of 16 in 16..512.  No reference counterpart. */
int32_t qp_group_rows(int32_t ldq);
int qp_schedule_gate_groups(int32_t T, const int32_t* slice_edge_ptr, const uint16_t* eu, const uint16_t* ev,
                            int32_t ldq, int32_t* group_ptr_out /* [ceil(T/G)+1] */, uint32_t* sched_out,
                            int64_t sched_capacity);

/* Chooses the device representation of a host distance matrix and writes it into a host staging
 * buffer the caller then copies to the device (C*C*8 bytes is always enough).  The diagonal is
 * forced to 0 (the reference never adds a term for equal cores).  Returns the kind in *kind_out. */
int qp_dtable_pack(const double* D_host, int32_t C, void* staging /* host */, size_t staging_bytes,
                   int32_t* kind_out, size_t* bytes_out);

/* Tests whether a host distance matrix is a grid metric (see QpGridMetric); *out is zeroed when not. */
int qp_dtable_grid_metric(const double* D_host, int32_t C, QpGridMetric* out);

/* The same test for a real-valued matrix that is one positive real number times an integer grid metric:
 * s = the smallest positive entry, M = D / s rounded to integers; accepted when |D[a][b] - s * M[a][b]| <=
 * rel_tol * D[a][b] for every pair and M passes qp_dtable_grid_metric.  *out describes M and *scale_out = s;
 * both are zeroed when the matrix is not of this form (or is an integer matrix already: scale 1 is
 * qp_dtable_grid_metric's case).  The distances the reference adds up are sums of k equal link weights formed
 * one addition at a time (Placer.py:63-76), so they differ from k * s by rounding only; a cost formed as
 * s * (number of hops) differs from the reference's sum by that much per term. */
int qp_dtable_scaled_grid_metric(const double* D_host, int32_t C, double rel_tol, QpGridMetric* out, double* scale_out);

/* 1 if |D[a][b] - D[b][a]| <= rel_tol * max(|D[a][b]|, |D[b][a]|) for every pair (infinite entries must
 * match exactly, NaN never passes), else 0; negative on bad arguments.  Distances over an undirected graph
 * differ from their transposes only by the rounding of the sums along the path (Placer.py:63-76 accumulates
 * from the source), a few units in the last place. */
int qp_dtable_symmetric(const double* D_host, int32_t C, double rel_tol);

/* HQA / TRHQA "make space" step for the listed instances of a batch (HQA_placer.py:180-204 ==
 * TRHQA_placer.py's copy): in core order, every core that holds more than cap-2 qubits gives up to two of
 * them (lowest qubit index first, as the row stands when the core's turn comes; stops as soon as the core
 * is down to cap-2) to the core with room (fewer than cap-1 qubits) that is nearest by D[core][.], ties by
 * lower occupancy, then lower core index.  a: host uint8 [.][lda] assignment rows, occ: host int64 [.][C]
 * core occupancies, both updated in place for rows[0..n_rows); D: host double [C][C]. */
int qp_make_space(uint8_t* a, int64_t* occ, const int64_t* rows, int64_t n_rows, int32_t Q, int32_t lda,
                  int32_t C, int32_t cap, const double* D);

/* n_rep placements by the RANDOM placer (Random_placer.py:25-59: per slice every active qubit draws
 * random.choice over the cores that still have room) or the BFS placer (BFS_placer.py:101-200: two-qubit gates to
 * the first core with two free slots, else to the closest pair of neighbouring cores with one free slot each;
 * one-qubit gates draw random.choice; the rest is spread breadth-first from the emptiest cores, BFS_placer.py:7-18
 * ordering the neighbours), consuming Python's random stream exactly as those loops do.
 * member_ptr [T+1] / members: the active qubits of every slice in the order the reference iterates its set of them;
 * cmd_ptr [T+1] / cmd_q0 / cmd_q1 (BFS): the commands of every slice in slice order, q1 = -1 for a one-qubit
 * command, q0 = -1 for a command of any other arity; nb_ptr [C+1] / nb (BFS): the neighbours of every core,
 * nearest first; D (BFS): host double [C][C].  Cores are numbered in the order of network.nodes().
 * mt_states: host uint32 [n_states][625], each the 624 words and the position of a Mersenne Twister as
 * random.getstate()[1] lists them, advanced in place; n_states = 1 (one stream through all repeats, as a loop of
 * calls would use it) or n_rep (repeat r draws from state r).
 * A_out: host uint8 [n_rep][T][ldq], QP_ABSENT where a qubit is not in the slice's dict, 0 in the padding. */
#define QP_PLACE_RANDOM 0
#define QP_PLACE_BFS 1
int qp_place_sparse(int32_t which, int32_t T, int32_t Q, int32_t ldq, int32_t C, int32_t cap,
                    const int32_t* member_ptr, const uint16_t* members, const int32_t* cmd_ptr,
                    const int32_t* cmd_q0, const int32_t* cmd_q1, const int32_t* nb_ptr, const int32_t* nb,
                    const double* D, uint32_t* mt_states, int32_t n_states, int32_t n_rep, uint8_t* A_out);

/* ---------------------------------------------------------------- device: cost evaluation */

/* Workspace (bytes, dev) qp_slice_cost_v3 / _v2 need for B partitions of circuit (T slices). */
size_t qp_slice_cost_workspace(int32_t T, int64_t B);

/* Total communication cost of B partitions of one circuit, HQA-family definition
 * (HQA_placer.py:259-287 == TRHQA_placer.py:551-585 == GTRHQA_placer.py:358-383):
 *   sum over slices i, counted gates (u,v) of slice i with A[i][u] != A[i][v]:  D[A[i][u]][A[i][v]]
 * + sum over i = 2..T-1, all qubits q with A[i][q] != A[i-1][q]:                 D[A[i-1][q]][A[i][q]]
 * A: dev uint8 [B][Tp][ldq] (Tp >= T rows per partition, row pitch ldq >= Q, ldq % 16 == 0); row i is
 * used for slice i whatever Tp is (GTRHQA returns T+1 rows).  out: dev double [B].
 * Sums are formed in a fixed order (bitwise reproducible run to run); exact when D is QP_DT_U16.  With
 * QP_DT_F64 and dt->symmetric set, a table too large for on-chip memory is applied as
 * (D[a][b] + D[b][a]) / 2 to the number of times the unordered pair {a, b} occurs. */
int qp_slice_cost_v3(const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B, int32_t Tp,
                     int32_t ldq, double* out, void* workspace, size_t workspace_bytes, void* stream);

/* The same sum slice by slice: out[b][t - t_begin] (dev double [B][t_end - t_begin]) = what timeslice t adds in
 * hqa_placer.per_timeslice_cost (HQA_placer.py:259-287: its crossing gates, :266-274, and from slice 2 on the
 * qubits that are on another core than in slice t - 1, :277-285), for t_begin <= t < t_end.  The rows of A may be
 * a window of the circuit: row r of a partition is the assignment of slice row0 + r, and the window must hold rows
 * t_begin - 1 (from slice 2 on) to t_end - 1.  A long circuit is costed in contiguous pieces this way, one per
 * GPU, with one row shared between neighbours; the pieces' sums add up to qp_slice_cost_v3's value to rounding.
 * Entries are read from dt->table (either kind); needs no workspace. */
int qp_slice_cost_v3_slices(const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B, int32_t Tp,
                            int32_t ldq, int32_t row0, int32_t t_begin, int32_t t_end, double* out, void* stream);

/* qp_slice_cost_v3 with one distance table per partition: tables is dev double [S][C*C] (row = source core,
 * zero diagonal, entries may be +inf), partition b is costed with table tab_index[b] (dev int32 [B], values
 * 0..S-1).  This is the cost step of simulate_broken_links (HQA_placer.py:325-361 and its copies in
 * TRHQA_placer.py:628-669, BFS_placer.py:241-278, Random_placer.py:255-296) for all damaged networks in one launch.
 * Needs no workspace; sums are formed in a fixed order. */
int qp_slice_cost_v3_multi(const QpCircuit* circ, const double* tables, int32_t S, int32_t C, const int32_t* tab_index,
                           const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, double* out, void* stream);

/* qp_slice_cost_v3 for callers that cannot allocate device memory or hand over a workspace: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * Same arguments, same kernels and bit-for-bit the same results as qp_slice_cost_v3; the partial sums (and
 * the symmetrised triangle of a large fp64 table) live in a QP_STATIC_SCRATCH_BYTES array inside the
 * library's code object, of which every GPU holds its own copy until the process ends.  The call allocates
 * and frees nothing.  A batch that needs more scratch (about one double per partition and chunk of 16..256
 * slices, plus 256 KB for an fp64 table of more than 134 cores) is refused with QP_ENOSPC: split it, or use
 * qp_slice_cost_v3.  Calls on one GPU share that GPU's array: two of them in flight at once on different
 * streams race (calls on one stream are ordered and safe; nothing has to be cleared between calls).
 * Under stream capture: make one ordinary call on the device first (it looks up the scratch array's address, which may
 * load the code object, and sets the kernels' shared-memory attributes); after that the call only queries occupancy and
 * enqueues kernels, and a captured graph of it can be replayed on new contents of A (tests/test_zz_no_alloc.py does so).
 * Which kernel runs depends on the circuit, the table and the batch only, so it is the one qp_slice_cost_v3 runs. */
#define QP_STATIC_SCRATCH_BYTES (4u << 20)
size_t qp_static_scratch_bytes(void);
int qp_slice_cost_v3_static(const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B, int32_t Tp,
                            int32_t ldq, double* out, void* stream);

/* Gate-ordered cost of B sparse partitions (QP_ABSENT = qubit not in the slice's dict).
 * flags selects the reference variant:
 *   QP_V2  BFS_placer.py:202-239 == Random_placer.py:178-213
 *   QP_V1  BasePlacer.per_timeslice_cost, Placer.py:78-120 (Tp may be < T: later slices are skipped)
 *   QP_V4  Qubo_placer.py:177-207 (pass the hop-count table; movement is looked up as [cur][last])
 * Where the reference would raise (absent qubit of a counted gate in V2/V4; a qubit placed at its
 * previous visit and absent now) out[b] = +inf, which is what main.py:99-102 records for the repeat.
 * With visit_mask (and visit_prev) in the circuit, batches of 64 MB or more on a circuit that visits an eighth
 * or more of every row (rows of up to 1024 bytes) are read row by row: a QP_DT_U16 table gives the same integers; a QP_DT_F64 table with
 * dt->symmetric set is applied as (D[a][b] + D[b][a]) / 2 to counts of core pairs (as in qp_slice_cost_v3; within
 * the table's asymmetry of the visit-by-visit sums).  Sums are formed in a fixed order either way. */
#define QP_V2 0
#define QP_V1 1
#define QP_V4 2
int qp_slice_cost_v2(const QpCircuit* circ, const QpDTable* dt, const uint8_t* A, int64_t B, int32_t Tp,
                     int32_t ldq, int32_t flags, double* out, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---------------------------------------------------------------- device: cost matrices */

/* One op x core cost matrix per instance, for the current while-iteration of the placement loop.
 * kind: QP_CM_HQA    HQA_placer.py:112-122 with :208-241
 *       QP_CM_TRHQA  TRHQA_placer.py:135-166 with :471-532
 *       QP_CM_GTRHQA GTRHQA_placer.py:99-112
 * t[b]: current timeslice; a: dev uint8 [B][ldq] live assignment; ops: dev int32 [B][max_ops] indices
 * into circ->edges; avail: dev uint8 [B][C] core ids (first n_avail[b] valid).
 * cost_out: dev double [B][max_ops][C], entry [b][i][j] valid for i < n_ops[b], j < n_avail[b].
 * The distance table must be QP_DT_F64.  Arithmetic is done one rounding at a time in the
 * reference's order (no FMA contraction), so entries are bit-identical to the reference's. */
#define QP_CM_HQA 0
#define QP_CM_TRHQA 1
#define QP_CM_GTRHQA 2
int qp_cost_matrix(int32_t kind, const QpCircuit* circ, const QpDTable* dt, int64_t B, const int32_t* t,
                   const uint8_t* a, int32_t ldq, const int32_t* ops, const int32_t* n_ops,
                   int32_t max_ops, const uint8_t* avail, const int32_t* n_avail, double* cost_out,
                   void* stream);

/* qp_cost_matrix with one distance table per instance: tables is dev double [S][C*C] (row = source core, zero
 * diagonal), instance b reads table tab_index[b] (dev or device-visible host int32 [B]).  For placements on many
 * damaged copies of one network at once (simulate_broken_links, HQA_placer.py:325-361, TRHQA_placer.py:628-669). */
int qp_cost_matrix_multi(int32_t kind, const QpCircuit* circ, const double* tables, int32_t S, int32_t C,
                         const int32_t* tab_index, int64_t B, const uint8_t* a, int32_t ldq, const int32_t* ops,
                         const int32_t* n_ops, int32_t max_ops, const uint8_t* avail, const int32_t* n_avail,
                         double* cost_out, void* stream);

/* ---------------------------------------------------------------- HQA / TRHQA placement batches */

/* B independent HQA (kind QP_CM_HQA) or TRHQA (QP_CM_TRHQA) placements of one circuit: the whole control loop
 * of hqa_placer.place_per_timeslice (HQA_placer.py:56-176: per slice the gates whose qubits sit on different
 * cores, the while-loop with its iteration budget, make-space when no core has room (:180-204), the op x core
 * cost matrix (:112-122 with :208-241; TRHQA_placer.py:135-166 with :471-532), linear_sum_assignment (:126) and
 * the moves (:129-145)).  Cost matrices are computed on the device (the kernel of qp_cost_matrix, fed with the
 * cores of every op's qubits and look-ahead partners as the host resolved them; circ->lookahead is copied to the
 * host once per call); the assignment solves stay on the host.  The batch is split over `lanes` host threads, each driving its instances
 * through all slices on its own stream: streams[l] is the HIP stream of lane l (lanes of them, all on the current
 * device, none of them busy with work this call must wait for).
 * circ / dt: device tables as for qp_cost_matrix (dt QP_DT_F64).  slice_edge_ptr [T+1], edges [nE], D [C*C]:
 * host copies of the circuit's gate list and of the distance matrix.  initial: host uint8 [B][Q], core index per
 * qubit at slice 0.  rows_out: host uint8 [B][T][Q], row t = assignment after slice t (row 0 = initial).
 * pinned: page-locked host memory the device can address (hipHostMalloc or equivalent), 16-byte aligned,
 * qp_hqa_place_pinned_bytes(ldq, C, most gates in a slice, B, lanes) bytes with ldq = Q rounded up to 16: the
 * kernel reads its inputs from it and writes the matrices into it, nothing is copied. */
size_t qp_hqa_place_pinned_bytes(int32_t ldq, int32_t C, int32_t max_slice_edges, int64_t B, int32_t lanes);
int qp_hqa_place_batch(int32_t kind, const QpCircuit* circ, const QpDTable* dt, const int32_t* slice_edge_ptr,
                       const QpEdge* edges, const double* D, int32_t cap, const uint8_t* initial, int64_t B,
                       uint8_t* rows_out, void* pinned, size_t pinned_bytes, int32_t lanes, void* const* streams);

/* The same with one distance table per instance (placements on many damaged copies of one network,
 * simulate_broken_links, HQA_placer.py:325-361 / TRHQA_placer.py:628-669): tables dev double [S][C*C], D host
 * double [S][C*C] (the same matrices), tab_index host int32 [B] (null: every instance uses table 0). */
int qp_hqa_place_batch_multi(int32_t kind, const QpCircuit* circ, const double* tables, int32_t S, int32_t C,
                             const int32_t* tab_index, const int32_t* slice_edge_ptr, const QpEdge* edges, const double* D,
                             int32_t cap, const uint8_t* initial, int64_t B, uint8_t* rows_out, void* pinned,
                             size_t pinned_bytes, int32_t lanes, void* const* streams);

/* ---------------------------------------------------------------- device: GTRHQA placement */

/* B independent GTRHQA placements of one circuit, every timeslice on the device (one workgroup walks
 * one instance; no host round trip per slice).  Replaces gtrhqa_placer.place_per_timeslice,
 * GTRHQA_placer.py:62-265: slice t starts from row t-1 (row 0 for t = 0) and its result is row t+1;
 * rows = the slice's counted gates whose qubits sit on different cores (slice order), columns = cores
 * holding at most cap-2 qubits when the slice starts (ascending); entries as QP_CM_GTRHQA; assignment by
 * the bidding rounds of :128-220 and the leftover pass of :222-254.
 * initial: dev uint8 [B][ldq] core index per qubit (< C; row 0 of every instance).  out: dev uint8
 * [B][T+1][ldq]; row 0 is a copy of initial (pad columns included), the rest follow.
 * dt must be QP_DT_F64.  The caller checks what the reference checks before it starts (capacity).
 * workspace: dev, qp_gtrhqa_workspace(C) bytes, 16-byte aligned (prepared copies of the table: fp64, and 16-bit
 * for tables of small integers, which the call detects itself; and for every pair of source cores the sixteen
 * cheapest cores of an op whose qubits sit on them, in bidding order, which most bids are read from). */
size_t qp_gtrhqa_workspace(int32_t C);
int qp_gtrhqa_place(const QpCircuit* circ, const QpDTable* dt, const uint8_t* initial, int64_t B, int32_t ldq,
                    int32_t cap, uint8_t* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- device: activity counts */

/* Inter-core events every core takes part in, per partition: hqa_placer.per_core_intercore_activity,
 * HQA_placer.py:288-321 (== TRHQA_placer.py:586-619).  A as for qp_slice_cost_v3 (dense);
 * out: dev uint64 [B][C], overwritten. */
int qp_core_activity(const QpCircuit* circ, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, int32_t C,
                     uint64_t* out, void* stream);

/* Symmetric core x core event counts of the gate-ordered walk: qubo_placer.intercorecommatrix,
 * Qubo_placer.py:208-245 (the counts, not the heat-map file).  A as for qp_slice_cost_v2 (sparse,
 * QP_ABSENT allowed).  out: dev int32 [B][C][C], overwritten; status: dev int32 [B], 1 where the
 * reference would raise (absent qubit of a counted gate, or a qubit placed before and absent now). */
int qp_comm_matrix(const QpCircuit* circ, const uint8_t* A, int64_t B, int32_t Tp, int32_t ldq, int32_t C,
                   int32_t* out, int32_t* status, void* stream);

/* ---------------------------------------------------------------- device: QUBO placer */

/* The QUBO placer's objective (Qubo_placer.py:84-133) over binary variables x[t,j,i] (active qubit i
 * of slice t on core j) and slack bits y[t,j,p], written as
 *     E(s) = constant + sum_v linear[v] s_v + sum_{v<w} S[v][w] s_v s_w,
 * all integers.  State vector: slice t occupies [off[t], off[t+1]); inside it v = off[t] + j*pitch[t] + k, k < n[t]
 * for x, n[t] <= k < n[t]+cap for y, the rest of the pitch (n[t]+cap rounded up to 32) and of the slice (rounded
 * up to 128) is padding that must be zero.  S is never stored.  Inside a slice it is the Kronecker sum
 * I_C (x) same[t] + (J-I)_C (x) diag(8 on the x rows): same[t] holds the couplings between the variables of one
 * core (x-x +8, -20 more for interacting qubits, x-y -8, y-y +8; the same for every core) and is the only dense
 * piece, an int8 matrix contracted with the states by MFMA, core by core.  Between slices t and t+1 it is
 * hops10 (x) matching(next[t]): 10 * hops[j][l] between x[t,j,i] and x[t+1,l,next[t][i]]. */
typedef struct QpQuboKron {
    int32_t T, C, cap, nmax;   /* slices, cores (<= 32), capacity (<= 32), max_t n[t] */
    int64_t Npad;              /* length of a state vector, multiple of 128 */
    int32_t constant;
    int32_t n_tiles;           /* sum_t C * (rows of same[t] / 128) */
    const int32_t* n;          /* dev [T] active qubits per slice */
    const int32_t* pitch;      /* dev [T] state bytes per core, multiple of 32 */
    const int64_t* off;        /* dev [T+1] */
    const int8_t* same;        /* dev: same[t] at same + same_off[t], row-major, rows = columns = pitch[t] rounded
                                  up to 128, zero outside the (n[t]+cap)^2 block; 16-byte aligned */
    const int64_t* same_off;   /* dev [T+1] */
    const int32_t* tile_slice; /* dev [n_tiles]: slice of every (core, 128-row tile of same[t]) work item */
    const int32_t* tile_first; /* dev [T+1]: first work item of every slice */
    const int32_t* linear;     /* dev [Npad] */
    const int32_t* next;       /* dev [T][nmax]: index in slice t+1 of the same qubit, -1 = not active there */
    const int32_t* prev;       /* dev [T][nmax]: the same towards slice t-1 */
    const int32_t* hops10;     /* dev [C*C]: 10 * hop count */
} QpQuboKron;

/* Structure of the same model for the annealer (all pointers dev; per-(slice, qubit) tables have row
 * pitch nmax = max_t n_t). */
typedef struct QpQuboStruct {
    int32_t T, C, cap, nmax;
    int64_t Npad;
    const int32_t* n;       /* [T] active qubits per slice */
    const int64_t* off;     /* [T+1] */
    const uint64_t* info;   /* [T*nmax] per (slice, active qubit): bits 0-7 interaction-graph degree; then
                               three 16-bit words t' * nmax + i' into the [T][nmax] table of the same
                               shape, T * nmax meaning "none": 16-31 the first interaction partner
                               (same slice), 32-47 the same qubit in slice t+1, 48-63 in slice t-1.
                               T * nmax must stay below 65535. */
    const int32_t* nbr_ptr; /* [T*nmax+1] CSR over interaction partners (used when the degree is > 1) */
    const int16_t* nbr;
    const uint8_t* hops;    /* [C*C] hop counts */
    const int32_t* pitch;   /* [T] state bytes per core (QpQuboKron.pitch) */
} QpQuboStruct;

/* Device workspace (bytes) of the two calls below for R replicas: one core mask per (slice, qubit, replica). */
size_t qp_qubo_workspace(const QpQuboKron* model, int32_t R);

/* Local fields F[v][r] = sum_w S[v][w] s_w of R replica states (int8 0/1, stored replica-major:
 * Xt[r][v], r < R, v < Npad; R a multiple of 128; 16-byte aligned).  F: dev int32 [Npad][R] (zero on padding).
 * The energy change of flipping v in replica r is (1 - 2 s_v)(linear[v] + F[v][r]). */
int qp_qubo_fields_i32(const QpQuboKron* model, const int8_t* Xt, int32_t R, int32_t* F, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Exact energies energy[r] = E(state r) of R replica states: the same contraction with the reduction
 * sum_v s_v (2 linear[v] + F[v][r]) fused into its epilogue (integer atomics: order-independent).
 * This is the number pyqubo's decoded sample reports as .energy (Qubo_placer.py:141-142). */
int qp_qubo_energy_i32(const QpQuboKron* model, const int8_t* Xt, int32_t R, int32_t* energy, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Replica-parallel simulated annealing (stands in for neal.SimulatedAnnealingSampler().sample,
 * Qubo_placer.py:139-140; R plays num_reads).  Metropolis sweeps over all variables with a geometric
 * inverse-temperature schedule beta_hot -> beta_cold.  Xt: dev int8 [R][Npad], final states on return
 * (initial states on entry when QP_QUBO_INIT_GIVEN, else random).  dE_out[r]: summed energy change of
 * the accepted flips of replica r.  Deterministic for a given seed. */
#define QP_QUBO_INIT_GIVEN 1
int qp_qubo_anneal(const QpQuboStruct* model, int32_t R, int32_t sweeps, double beta_hot, double beta_cold,
                   uint64_t seed, int8_t* Xt, int32_t* dE_out, int32_t flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QPLACE_H */
