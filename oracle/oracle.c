/*
 * oracle.c — CPU restatement ("oracle") of the mobvoi/lstm_ctc hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker / the
 * reported CPU baseline.  The product path (lstm_ctc_amd/) never links or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE (no tests/golden vectors exist in the reference and
 * TensorFlow 1.8 is not installable here).  Pinned by: TF-upstream CTC known answers,
 * torch-CPU cross-checks and fp64 finite differences — see oracle/README.md.
 *
 * Every function cites the reference file:line whose behaviour it follows; the arithmetic
 * itself lives in the un-vendored dependency tensorflow==1.8.0 (README.md:23 of the
 * reference), restated from its published algorithm.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define REAL float
#define FN(x) x##_f32
#define EXP expf
#define LOG logf
#define LOG1P log1pf
#define TANH tanhf
#include "oracle_core.inc"
#undef REAL
#undef FN
#undef EXP
#undef LOG
#undef LOG1P
#undef TANH

#define REAL double
#define FN(x) x##_f64
#define EXP exp
#define LOG log
#define LOG1P log1p
#define TANH tanh
#include "oracle_core.inc"

/*
 * tf.edit_distance(hyp, truth, normalize=False) — nnet/graph.py:143-149; TF r1.8
 * core/kernels/edit_distance_op.cc + lib/gtl/edit_distance.h:
 * Levenshtein distance with unit insert/delete/substitute cost, per utterance.
 * hyp [B,hyp_stride] with hyp_len[b] valid tokens, truth flat + offsets[B+1]; dist [B].
 */
void orc_edit_distance(const int *hyp, int hyp_stride, const int *hyp_len, const int *truth,
                       const int *offs, int B, int *dist)
{
    for (int b = 0; b < B; ++b) {
        const int *h = hyp + (size_t)b * hyp_stride, *r = truth + offs[b];
        int n = hyp_len[b], m = offs[b + 1] - offs[b];
        int *row = (int *)malloc(sizeof(int) * (m + 1));
        for (int j = 0; j <= m; ++j) row[j] = j;
        for (int i = 1; i <= n; ++i) {
            int diag = row[0];
            row[0] = i;
            for (int j = 1; j <= m; ++j) {
                int sub = diag + (h[i - 1] != r[j - 1]);
                int del = row[j] + 1, ins = row[j - 1] + 1;
                diag = row[j];
                int v = sub < del ? sub : del;
                row[j] = v < ins ? v : ins;
            }
        }
        dist[b] = row[m];
        free(row);
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* The host may expose far more logical CPUs than the process is allowed to use (container CPU quota): the caller
 * passes the usable count (oracle.py: min(affinity, cgroup quota)); spinning surplus threads cost 50x on such a box. */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
