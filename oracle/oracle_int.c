/* TEST INFRASTRUCTURE ONLY: plain-C restatement of the integer parts of the Allophant prediction path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product never does.
 *
 *   oracle_frame_lengths  floor((len - k)/s) + 1 per conv layer -- allophant/network/frontend.py:192-203 applied as in
 *                         allophant/network/acoustic_model.py:832-835 (torch.div rounding_mode="floor")
 *   oracle_greedy_ctc     argmax / unique_consecutive / drop blank / 1-based start timesteps / score
 *                         -- allophant/predictions.py:194-207
 * Pinned by tests/test_oracle_golden.py against tests/golden/g4_integer.npz (outputs of the reference itself).
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/liboracle_int.so oracle/oracle_int.c
 */
#include <stdint.h>

static int64_t floor_div(int64_t a, int64_t b) {
    int64_t q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

void oracle_frame_lengths(const int64_t* lengths, int n, const int32_t* kernels, const int32_t* strides, int n_conv,
                          int64_t* out) {
    for (int i = 0; i < n; ++i) {
        int64_t len = lengths[i];
        for (int c = 0; c < n_conv; ++c) len = floor_div(len - kernels[c], strides[c]) + 1;
        out[i] = len;
    }
}

/* log_emissions: [N, T, C] batch-major (the layout the reference decoder receives, run.py:767-774);
 * tokens/timesteps: [N, T] (first counts[n] entries valid). */
void oracle_greedy_ctc(const float* log_emissions, const int64_t* lengths, int n, int t_max, int c, int64_t blank,
                       int64_t* tokens, int64_t* timesteps, int32_t* counts, double* scores) {
    for (int u = 0; u < n; ++u) {
        const float* e = log_emissions + (int64_t)u * t_max * c;
        int64_t prev = -1;
        int32_t k = 0;
        double score = 0.0;
        for (int64_t t = 0; t < lengths[u]; ++t) {
            const float* row = e + t * c;
            int64_t best = 0;
            for (int j = 1; j < c; ++j)
                if (row[j] > row[best]) best = j;
            score += row[best];
            if (t == 0 || best != prev) {
                if (best != blank) {
                    tokens[(int64_t)u * t_max + k] = best;
                    timesteps[(int64_t)u * t_max + k] = t + 1;
                    ++k;
                }
            }
            prev = best;
        }
        counts[u] = k;
        scores[u] = score;
    }
}
