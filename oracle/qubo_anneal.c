/* TEST INFRASTRUCTURE ONLY -- CPU baseline of the QUBO placer's sampler, never linked into the product.
 *
 * A plain sequential single-flip Metropolis annealer over a sparse QUBO, the textbook form of what
 * neal.SimulatedAnnealingSampler().sample does at Qubo_placer.py:139-140 (neal itself is absent here and unpinned
 * in the reference): per sweep every variable in index order, energy change from a running local field, on
 * acceptance the fields of the variable's neighbours are updated; geometric inverse-temperature schedule.
 * E(s) = const + sum_v lin[v] s_v + sum_{v<w} S[v][w] s_v s_w with S given as a symmetric CSR (both directions). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline uint64_t next64(uint64_t* s) { /* xorshift64* */
    uint64_t x = *s;
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1DULL;
}

/* state: in = initial 0/1 bytes, out = final.  Returns the summed energy change of the accepted flips;
 * *accepted_out counts them.  Returns INT64_MIN when memory runs out. */
int64_t oracle_qubo_anneal(int32_t N, const int64_t* rowptr, const int32_t* col, const int32_t* val, const int32_t* lin,
                           int32_t sweeps, double beta_hot, double beta_cold, uint64_t seed, int8_t* state,
                           int64_t* accepted_out) {
    int64_t* field = (int64_t*)calloc((size_t)(N > 0 ? N : 1), sizeof(int64_t));
    if (!field) return INT64_MIN;
    for (int32_t v = 0; v < N; ++v)
        if (state[v])
            for (int64_t k = rowptr[v]; k < rowptr[v + 1]; ++k) field[col[k]] += val[k];
    uint64_t rng = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
    if (!rng) rng = 1;
    int64_t total = 0, accepted = 0;
    for (int32_t sw = 0; sw < sweeps; ++sw) {
        const double frac = sweeps > 1 ? (double)sw / (double)(sweeps - 1) : 0.0;
        const double beta = beta_hot * pow(beta_cold / beta_hot, frac);
        for (int32_t v = 0; v < N; ++v) {
            const int d = 1 - 2 * (int)state[v];
            const int64_t dE = (int64_t)d * ((int64_t)lin[v] + field[v]);
            int take = dE <= 0;
            if (!take) {
                const double u = (double)(next64(&rng) >> 11) * (1.0 / 9007199254740992.0);
                take = u < exp(-beta * (double)dE);
            }
            if (take) {
                state[v] = (int8_t)(state[v] + d);
                for (int64_t k = rowptr[v]; k < rowptr[v + 1]; ++k) field[col[k]] += (int64_t)d * val[k];
                total += dE;
                ++accepted;
            }
        }
    }
    free(field);
    if (accepted_out) *accepted_out = accepted;
    return total;
}

/* E(s) - const of one state, from the same CSR (every pair is stored twice). */
int64_t oracle_qubo_energy(int32_t N, const int64_t* rowptr, const int32_t* col, const int32_t* val, const int32_t* lin,
                           const int8_t* state) {
    int64_t e = 0, pairs = 0;
    for (int32_t v = 0; v < N; ++v) {
        if (!state[v]) continue;
        e += lin[v];
        for (int64_t k = rowptr[v]; k < rowptr[v + 1]; ++k)
            if (state[col[k]]) pairs += val[k];
    }
    return e + pairs / 2;
}
