/* TEST INFRASTRUCTURE ONLY — C restatement of the reference's communication-cost loops, used by
 * tests/ and by bench.py's cpu_baseline leg; nothing in the product links or loads it.
 *
 * Sums are formed left to right in the reference's order, so with the same distance matrix the
 * result has the same bits as the reference's Python float (checked against tests/golden/).
 *
 * Data: gates in slice order as (q0, q1, counted) triples with a CSR over slices; a partition is
 * uint8 rows of pitch ldq (0xFF = qubit absent from the slice's dict); D is row-major doubles.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* HQA_placer.py:259-287 (== TRHQA_placer.py:551-585, GTRHQA_placer.py:358-383). */
double oracle_cost_v3(int Q, int T, const int32_t* slice_ptr, const int32_t* gq0, const int32_t* gq1,
                      const uint8_t* counted, const uint8_t* A, int ldq, const double* D, int C) {
    double total = 0.0;
    for (int i = 0; i < T; ++i) {
        const uint8_t* row = A + (size_t)i * ldq;
        for (int g = slice_ptr[i]; g < slice_ptr[i + 1]; ++g) {
            if (!counted[g]) continue;
            const int a = row[gq0[g]], b = row[gq1[g]];
            if (a != b) total += D[(size_t)a * C + b];
        }
    }
    for (int i = 2; i < T; ++i) {
        const uint8_t* prev = A + (size_t)(i - 1) * ldq;
        const uint8_t* cur = A + (size_t)i * ldq;
        for (int q = 0; q < Q; ++q)
            if (cur[q] != prev[q]) total += D[(size_t)prev[q] * C + cur[q]];
    }
    return total;
}

/* BFS_placer.py:202-239 / Random_placer.py:178-213 (mode 0), Placer.py:78-120 (mode 1),
 * Qubo_placer.py:177-207 (mode 2, pass hop counts).  Returns +inf where the reference raises.
 * last[] is caller scratch of Q ints. */
double oracle_cost_v2(int Q, int T, int Tp, const int32_t* slice_ptr, const int32_t* gq0, const int32_t* gq1,
                      const uint8_t* counted, const uint8_t* A, int ldq, const double* D, int C, int mode,
                      int32_t* last) {
    double total = 0.0;
    for (int q = 0; q < Q; ++q) last[q] = -1;
    for (int i = 0; i < T; ++i) {
        if (i >= Tp) {
            if (mode == 1) break;
            return INFINITY;
        }
        const uint8_t* row = A + (size_t)i * ldq;
        for (int g = slice_ptr[i]; g < slice_ptr[i + 1]; ++g) {
            const int qs[2] = {gq0[g], gq1[g]};
            const int nq = qs[1] < 0 ? 1 : 2;
            if (counted[g]) {
                const int a = row[qs[0]], b = row[qs[1]];
                if (a == 0xFF || b == 0xFF) {
                    if (mode != 1) return INFINITY;
                } else if (a != b) {
                    total += D[(size_t)a * C + b];
                }
            }
            for (int k = 0; k < nq; ++k) {
                const int q = qs[k];
                const int cur = row[q] == 0xFF ? -1 : row[q];
                if (last[q] >= 0 && cur != last[q]) {
                    if (cur < 0) return INFINITY;
                    total += mode == 2 ? D[(size_t)cur * C + last[q]] : D[(size_t)last[q] * C + cur];
                }
                if (mode == 1) {
                    if (cur >= 0) last[q] = cur;
                } else {
                    last[q] = cur;
                }
            }
        }
    }
    return total;
}
