/*
 * lstm_ctc_hip.h — C ABI of liblstm_ctc_hip.so, the MI355X (gfx950) implementation of the
 * mobvoi/lstm_ctc training/inference hot path.
 *
 * The reference has NO plugin/FFI boundary (pure Python 2 + TensorFlow 1.8);
 * each entry point below replaces the TensorFlow op(s) the reference calls at the cited
 * file:line.  A maintainer binds these with ctypes from the nnet package.
 *
 * Conventions
 *   - plain C types; every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *   - the library allocates nothing the caller can see (workspaces are sized by *_workspace_bytes);
 *   - every launch goes to the caller's hipStream_t (passed as void*), asynchronously;
 *   - return 0 on success, negative LC_E* on failure; lc_last_error() gives a thread-local message;
 *   - activations are TIME-MAJOR [T,B,*] float32 (the layout tf.nn.ctc_loss consumes at
 *     nnet/graph.py:72), labels are flat int32 + offsets[B+1] (the SparseTensor of graph.py:76-104).
 */
#ifndef LSTM_CTC_HIP_H
#define LSTM_CTC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_OK 0
#define LC_EINVAL (-1)   /* bad argument / unsupported shape */
#define LC_ELAUNCH (-2)  /* HIP launch or runtime failure */
#define LC_EWORKSPACE (-3)

typedef void *lc_stream_t; /* hipStream_t */

const char *lc_last_error(void);
int lc_version(void);

/* Development switches.  Each has an LC_* environment variable (read on every call) and a PER-THREAD override that
 * wins over it; LC_OPTION_UNSET clears the override.  lc_get_option returns the effective value (override, else
 * environment, else LC_OPTION_UNSET = "library default").  Names / variables:
 *   "lstm_persistent"  LC_LSTM_PERSISTENT   0 = every recurrence runs the per-step launch train
 *   "lstm_spin_limit"  LC_LSTM_SPIN_LIMIT   bound of the persistent kernels' waits, in polls
 *   "gemm_f32_big"     LC_GEMM_F32_BIG      0 = never the 256 x 256 LDS-DMA kernel, 2 = whenever eligible
 *   "gemm_bf16_big"    LC_GEMM_BF16_BIG     0 = never the 256 x 256 bf16 kernel
 *   "ctc_lse2"         LC_CTC_LSE2          lc_ctc_loss with a gradient: 1 = the per-frame log-sum-exp is taken by phase 2's
 *                                           frame waves (one logits read less; default from 512 utterances), 0 = by phase 1's
 * No reference counterpart (the reference has no native code). */
#define LC_OPTION_UNSET (-0x7fffffffL - 1)
int lc_set_option(const char *name, long value);
int lc_get_option(const char *name, long *value);

/* ------------------------------------------------------------------ CTC --------------------- */
/* tf.nn.ctc_loss(labels, inputs, sequence_length, ignore_longer_outputs_than_inputs=True)
 * as called at nnet/graph.py:109-114 (blank = V-1, ctc_merge_repeated=True).
 *   logits [T,B,V]; labels flat int32; label_offsets [B+1]; seq_len [B];
 *   max_label_len: host-side max_b (offsets[b+1]-offsets[b]);
 *   loss [B] (= -log p; 0 for skipped utterances, +inf when no valid path);
 *   grad [T,B,V] or NULL (d sum(loss) / d logits; 0 beyond seq_len). */
size_t lc_ctc_workspace_bytes(int T, int B, int V, int max_label_len);
int lc_ctc_loss(const float *logits, int T, int B, int V, const int *labels,
                const int *label_offsets, const int *seq_len, int max_label_len, float *loss,
                float *grad, void *workspace, size_t workspace_bytes, lc_stream_t stream);

/* tf.nn.ctc_greedy_decoder(inputs, sequence_length, merge_repeated=True) — nnet/graph.py:138-142.
 *   tokens [B,T] int32 (row b holds out_len[b] tokens), out_len [B] int32,
 *   workspace: T*B int32 (per-frame argmax). */
int lc_ctc_greedy(const float *logits, int T, int B, int V, const int *seq_len, int *tokens,
                  int *out_len, int *argmax_workspace, lc_stream_t stream);

/* tf.edit_distance(hyp, truth, normalize=False) — nnet/graph.py:143-149.  HOST function on HOST
 * buffers (integer DP on a few hundred tokens; SURVEY.md §2a keeps it on the host). */
int lc_edit_distance_host(const int *hyp, int hyp_stride, const int *hyp_len, const int *truth,
                          const int *truth_offsets, int B, int *dist);

/* ------------------------------------------------------------------ GEMM -------------------- */
/* C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C (+ bias[N] broadcast over rows if non-NULL),
 * float32 on the f32 MFMA pipe.  Row-major; ta/tb != 0 means the stored matrix is the transpose
 * (A stored [K,M], B stored [N,K]).  Replaces tf.matmul / tf.nn.xw_plus_b (nnet/bilstm.py:249,
 * nnet/moe.py:43,58) and the batched halves of the LSTMCell kernel matmul (bilstm.py:129-136). */
/* Tall-K products with few output tiles (weight gradients, K = T*B) are split along K into slabs in the
 * caller's workspace (lc_gemm_workspace_bytes; 0 = no split) and reduced deterministically.  A NULL or
 * too-small workspace just disables the split. */
size_t lc_gemm_workspace_bytes(int M, int N, int K);
int lc_gemm_f32(int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda,
                const float *B, int ldb, float beta, float *C, int ldc, const float *bias,
                void *workspace, size_t workspace_bytes, lc_stream_t stream);
/* Same product with bf16 OPERANDS (BASELINE.json configs[4], "bf16 MFMA gate GEMMs, fp32 CTC"): A and B stay
 * float32 in memory and are rounded to bf16 (round-to-nearest-even) as they are loaded,
 *   C = alpha * sum_k bf16(A[m,k]) * bf16(B[k,n]) + beta*C + bias,  accumulated in float32 by
 * v_mfma_f32_32x32x16_bf16.  Outputs, bias, alpha/beta arithmetic stay float32.  Same workspace rule. */
int lc_gemm_bf16(int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda,
                 const float *B, int ldb, float beta, float *C, int ldc, const float *bias,
                 void *workspace, size_t workspace_bytes, lc_stream_t stream);

/* bf16 SHADOW operands (second stage of the c5 path): the activations / weights are first copied to bf16 -
 * lc_cast_bf16 writes the same-orientation copy nat[rows][ldnat] and / or the transposed copy tr[C][ldtr] (either
 * may be NULL) of a float32 x[rows, C] (round-to-nearest-even) - and every product is then taken in "NT" form,
 *   C[M,N] = alpha * sum_k A[m,k] * B[n,k] + beta*C + bias,   A[M][K], B[N][K] bf16 with k contiguous,
 * so the loader is a plain 16-byte copy (no conversion, no transposition).  Bit-identical operands to lc_gemm_bf16
 * (same rounding), float32 accumulate / outputs.  K, lda, ldb multiples of 8; A, B 16-byte aligned. */
int lc_cast_bf16(const float *x, int rows, int C, int ldx, uint16_t *nat, int ldnat, uint16_t *tr, int ldtr,
                 lc_stream_t stream);
int lc_gemm_bf16_nt(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                    float beta, float *C, int ldc, const float *bias, void *workspace, size_t workspace_bytes,
                    lc_stream_t stream);
/* Two operand pairs into one result, C = alpha * (A1 B1^T + A2 B2^T) + beta*C + bias, in ONE pass over C (whole 256 x 256
 * tiles: one kernel whose reduction walks K1 then K2; elsewhere the two products one after the other): the input gradient
 * of a bidirectional layer - both cells read the same concatenated input, nnet/bilstm.py:190-203, so TF's gradient pass
 * adds the two tf.matmul(dz, Kx, transpose_b=True) nodes of nnet/bilstm.py:129-136,150-157.  Same operand rules as
 * lc_gemm_bf16_nt; both pairs share lda / ldb.  A fused epilogue (lc_gemm_next_epilogue) applies to the final value. */
int lc_gemm_bf16_nt2(int M, int N, int K1, int K2, float alpha, const uint16_t *A1, const uint16_t *A2, int lda,
                     const uint16_t *B1, const uint16_t *B2, int ldb, float beta, float *C, int ldc, const float *bias,
                     lc_stream_t stream);
/* The same two-pair product in float32 (config c4): A1, A2 [M][K] and B1, B2 [N][K] float32, k contiguous, 16-byte aligned
 * with lda / ldb multiples of 4 for the one-kernel route (K1, K2 multiples of 32, whole 256 x 256 tiles); anything else runs
 * the two lc_gemm_f32 products in sequence.  7.43 + 7.88 ms -> 14.78 ms at c4's dX shape (the second product's beta = 1 pass). */
int lc_gemm_f32_nt2(int M, int N, int K1, int K2, float alpha, const float *A1, const float *A2, int lda,
                    const float *B1, const float *B2, int ldb, float beta, float *C, int ldc, const float *bias,
                    lc_stream_t stream);
/* The same product with BOTH operands K-MAJOR: A stored [K][M], B stored [K][N] (C = alpha * A^T B + beta * C + bias) -
 * the weight gradients X^T dZ of a train step on the NATURAL bf16 shadows of the activations (their rows are the
 * reduction index), so no transposed copy is made.  M and N must be multiples of 256, lda / ldb multiples of 8, any K.
 * Replaces the tf.matmul(..., transpose_a=True) nodes TF's gradient pass builds for nnet/bilstm.py:129-136,249. */
int lc_gemm_bf16_tn(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                    float beta, float *C, int ldc, const float *bias, void *workspace, size_t workspace_bytes,
                    lc_stream_t stream);
/* ... and with A k-contiguous [M][K], B K-MAJOR [K][N] (C = alpha * A B + ...): the forward products X . Kx, hs . proj on
 * the natural shadows of activation and weight - no transposed weight copy.  M, N multiples of 256, K of 64. */
int lc_gemm_bf16_nn(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                    float beta, float *C, int ldc, const float *bias, void *workspace, size_t workspace_bytes,
                    lc_stream_t stream);

/* One-shot fused epilogue: arms the NEXT lc_gemm_* call of the calling thread (any of the five above; it is consumed by that
 * call whether it succeeds or not) to finish its output element (r, c) of the whole M x N result in the product kernel:
 *   keep < 1:  C[r][c] *= the DropoutWrapper factor of (seed, stream0 + c / drop_width, r * drop_width + c % drop_width) -
 *              bit-identical to lc_dropout_scale(keep, seed, stream0 + d) on each column window d of width drop_width
 *              (nnet/bilstm.py:147-160: one wrapper per direction; its backward applies the same mask to dY);
 *   c_bf16:    the final value, rounded to bf16 (RNE), also goes to c_bf16[r * ldc_bf16 + c] - the shadow operand the next
 *              product reads (what lc_cast_bf16 / lc_dropout_scale_bf16 would write in a pass of their own).
 * An armed product is never split along K (the mask lives in the product kernel), and beta, bias are applied BEFORE the mask.
 * e == NULL disarms.  Replaces the separate tf.nn.dropout node after each direction's projection. */
typedef struct lc_gemm_epilogue {
    float keep;            /* 1.0: no mask */
    uint32_t seed, stream0;
    int drop_width;        /* P: columns per dropout stream */
    uint16_t *c_bf16;      /* or NULL */
    int ldc_bf16;
    int shadow_only;       /* non-zero (needs c_bf16): ONLY the shadow is written; the float32 C is UNSPECIFIED afterwards (left
                            * untouched, except where lc_gemm_bf16_nt2 runs its two products in sequence - ragged edges, small
                            * shapes - and keeps the first one's partial sum there) - for
                            * outputs every later product reads through the shadow (a c5 step's layer outputs and input
                            * gradients): a third of the bytes, and of the store burst that ends every tile */
} lc_gemm_epilogue_t;
int lc_gemm_next_epilogue(const lc_gemm_epilogue_t *e);

/* fp32 products on the bf16 matrix cores ("bf16x3", config key compute_dtype = bf16x3): each fp32 operand element is split
 * exactly into three bf16 terms (hi + mid + lo) by lc_split_bf16x3, and lc_gemm_bf16x3_nt accumulates, in fp32, the six term
 * pairs of weight >= 2^-16 (what is dropped is <= 2^-25 of each product - a quarter of an fp32 ulp).  Same tf.matmul nodes as
 * lc_gemm_f32 (nnet/bilstm.py:129-136,249), fp32-grade results, 6 bf16 MFMAs per 16 k instead of 8 fp32 MFMAs.
 *
 * Range: the split is exact for 1e-30 <= |x| <= 3.38e38 (bf16's largest finite value) and for zero; smaller magnitudes lose
 * their low terms to the fp32 denormal flush (absolute error < 2^-126); Inf, NaN and |x| above bf16's maximum give NaN.
 *
 * Error bound (lc_gemm_bf16x3_nt, _tn, and the split-operand recurrences lc_lstm_fwd_x3 / _bwd_x3), per output element with
 * alpha = 1, beta = 0:
 *     | C - sum_k a_k b_k |  <=  3e-7 * max(1, sqrt(K) / 4) * sum_k |a_k| |b_k|  +  2^-125 * sum_k (|a_k| + |b_k|)
 * The first term is the fp32 accumulation both this product and the fp32 MFMA kernels carry (the dropped term pairs add
 * <= 2^-23 of it); the second is the flush of split terms below 2^-126 and only shows where an operand below ~1e-30 meets one
 * above ~1e+8.  Held by tests/test_gpu_ops.py::test_gemm_bf16x3_adversarial_operands on rows that cancel to 1e-6 of their
 * magnitude sum, per-row magnitude spreads of 2^24, operands in [1e-38, 1e-30], +-0 and fp32 denormals, next to the fp32
 * kernel on the same operands.  At MODEL level the claim is about FLOAT64 TRUTH, not about the fp32 mode: the split-operand
 * mode's logits are as far from a float64 evaluation of the same network as the fp32 kernels' are - measured at the benched
 * sizes with the reference's initialisation (profiles/r5_x3_truth.txt: rms error bf16x3 / fp32 = 0.97 .. 1.07 over
 * N = 320 .. 1024, 1 .. 5 layers, T = 300 .. 1000; tests/test_gpu_truth.py asserts <= 1.5 at N = 128, 512, 768, 1024 with
 * the schedule taken), and per step (profiles/r5_x3_local_error.txt: teacher-forced |dc|, |dh| over a T = 1000 trajectory
 * equal to the fp32 kernel's, 0.3 ulp of c).  The two modes do NOT agree with EACH OTHER on long sequences, and no two fp32
 * implementations do: at forget bias 5 the cell is a 150-step integrator (|c| up to ~650) and a 5-layer, T = 1000 stack
 * amplifies rounding-level differences until fp32 logits are uncorrelated with float64's (rms 0.52 on logits of rms 0.70
 * for the fp32 kernels, 0.52 for bf16x3, 0.45 for an independent fp32 implementation in torch, 0.78 for plain bf16) - so
 * `bf16x3 - fp32` (rms 0.42 at c4's full size) measures the workload's conditioning, not either mode.
 *
 * x3 shadow layout: row-major, row r = [k tile 0: hi[16] mid[16] lo[16] | k tile 1: ... ], K padded with zeros to a multiple
 * of 16; ldo (bf16 elements) >= 3 * roundup(cols, 16), multiple of 8; out 16-byte aligned. */
int lc_split_bf16x3(const float *x, int rows, int cols, int ldx, uint16_t *out, int ldo, lc_stream_t stream);
/* C[M,N] = alpha * A[M,K] * B[N,K]^T + beta * C + bias[N] on x3 shadows (both operands k-contiguous; any M, N, K).  Honours
 * lc_gemm_next_epilogue. */
int lc_gemm_bf16x3_nt(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                      float beta, float *C, int ldc, const float *bias, lc_stream_t stream);
/* C[M,N] = alpha * A^T B + beta * C + bias[N] with BOTH operands K-major x3 shadows - A of X [K, M], B of dZ [K, N]: the
 * weight gradients (tf.matmul(..., transpose_a=True) in TF's gradient of nnet/bilstm.py:129-136,249) on the SAME shadows the
 * forward / dX products read as row operands (row windows are fine: dR = hs_prev^T dZ).  Any M, N, K; with a workspace of
 * lc_gemm_bf16x3_tn_workspace_bytes the reduction is split along K (deterministic slices + a reduction pass) so that few
 * output tiles still fill the chip.  A pending lc_gemm_next_epilogue is consumed and ignored. */
size_t lc_gemm_bf16x3_tn_workspace_bytes(int M, int N, int K);
/* The same for operands with the given leading dimensions (row windows of wider shadows): a K slice of an operand is addressed
 * through one buffer descriptor (2 GB reach), so K * max(lda, ldb) * 2 bytes beyond that force more slices - and a workspace -
 * whatever the tile count.  lc_gemm_bf16x3_tn_workspace_bytes(M, N, K) is this with the minimal leading dimensions. */
size_t lc_gemm_bf16x3_tn_workspace_bytes_ld(int M, int N, int K, int lda, int ldb);
int lc_gemm_bf16x3_tn(int M, int N, int K, float alpha, const uint16_t *A, int lda, const uint16_t *B, int ldb,
                      float beta, float *C, int ldc, const float *bias, void *workspace, size_t workspace_bytes,
                      lc_stream_t stream);

/* ------------------------------------------------------------------ LSTM -------------------- */
/* The sequential part of tf.contrib.rnn.LSTMCell under tf.nn.dynamic_rnn with sequence_length
 * masking (nnet/bilstm.py:125-188), for one direction or for both directions
 * of a BiLSTM layer in the same launches.
 *
 * Column layout of every [.,4N] operand is GATE-INTERLEAVED: column (n/8)*32 + g*8 + n%8 holds gate
 * g (0=i,1=j,2=f,3=o) of unit n (TF keeps [i|j|f|o] blocks of N; the host permutes at checkpoint I/O).
 *   zx     [T,B,4N]  in: x_t.Kx + bias; out: activated gates (i, tanh j, f, o)
 *   R      [N,4N]    recurrent weights acting on m' (= proj.Kh with a projection layer, else Kh)
 *   w_f,w_i,w_o [N]  peepholes or NULL
 *   cs, hs [T,B,N]   out: cell state c_t and pre-projection output m'_t (0 where t >= seq_len[b])
 *   reverse != 0     run t = T-1..0: the reference's reverse_sequence -> dynamic_rnn -> reverse_sequence
 *                    (bilstm.py:112,180-190) done by index arithmetic.
 * Requires N % 8 == 0 (N % 16 == 0 for the backward). */
typedef struct {
    float *zx;
    const float *R;
    const float *w_f, *w_i, *w_o;
    float *cs, *hs;
    int reverse;
    uint16_t *hs_bf16;   /* lc_lstm_fwd_bf16 only, may be NULL: [T,B,N] bf16 (round-to-nearest-even) copy of hs, written in the
                          * same pass - the shadow operand the next product (m = m'.proj) reads, without a separate cast */
    int shadow_only;     /* lc_lstm_fwd_bf16 with hs_bf16: non-zero = the caller reads hs only through hs_bf16; the float32 hs
                          * is then UNSPECIFIED after the call (the full-width persistent kernel does not store it: one
                          * vector-memory instruction less per step).  zx and cs are written as always. */
} lc_lstm_fwd_dir_t;
/* Stream semantics: everything is ordered after prior work on `stream` and before later work on it.  For the big
 * bidirectional float32 case the reverse direction runs on an internal second stream that is forked from and joined
 * back into `stream` with events inside the call.
 * Schedules (same results up to float32 summation order): num_neurons <= 512, float32 and at most 64 batch rows
 * (128 uni-directional) run as ONE persistent launch per call - one XCD per (direction, row group), weights resident
 * in registers, state exchanged through that XCD's L2 (environment LC_LSTM_PERSISTENT=0 disables it); num_neurons in
 * {640, 768, 896, 1024} in float32 runs persistent launches on XCD PAIRS (the recurrent weights of half the units in each XCD's
 * registers, partial sums handed across): 64 batch rows of both directions - or 128 rows of one direction - per launch,
 * larger batches as consecutive launches over 64-row blocks of the same tensors, as long as T * B * 4N * 4 bytes < 2^32;
 * everything else runs one launch per time step.  The persistent launch needs the GPU's CUs free to become co-resident; every wait in
 * it is bounded, and a timeout writes NaN into the outputs instead of hanging. */
/* Failure reporting of the persistent schedule.  The first 256 bytes of the LSTM workspace are a control block; the
 * 32-bit word at byte LC_LSTM_STATUS_OFFSET is a STICKY status word: the library never clears it, and sets it to a
 * non-zero value when a persistent launch could not complete (a bounded wait ran out - e.g. its workgroups could not
 * become co-resident - or an XCD took no workgroups).  In that case every output row of the affected call is filled
 * with NaN.  The caller zeroes the word, runs any number of lc_lstm_* calls on the same workspace, and reads it at its
 * next synchronisation point (lc_optimizer_step can take it as `guard` so that a failed step leaves the parameters
 * untouched); on failure it re-runs the step with LC_LSTM_PERSISTENT=0.  The persistent schedule is only chosen on a
 * device that reports gfx950 with 256 CUs in 8 XCCs (SPX); anything else runs the launch train.
 * Environment (read per call; for tests): LC_LSTM_PERSISTENT=0 forces the launch train, LC_LSTM_SPIN_LIMIT=<n> bounds
 * every wait to n polls (default 2^21). */
#define LC_LSTM_STATUS_OFFSET 64
size_t lc_lstm_fwd_workspace_bytes(int B, int N, int ndir);
int lc_lstm_fwd(const lc_lstm_fwd_dir_t *dirs /* host array */, int ndir, const int *seq_len, int T, int B,
                int N, float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream);

/* BPTT of lc_lstm_fwd.
 *   gates [T,B,4N] in: activated gates; out: dz (gradient w.r.t. the pre-activations)
 *   RT    [4N,N]   transpose of R
 *   dh    [T,B,N]  gradient w.r.t. m'_t coming from the layer output
 *   dpeep [3,N]    += gradients of (w_f, w_i, w_o); may be NULL
 *   dbias [4N]     += column sums of dz = gradient of the LSTM bias (same column layout as gates); may be NULL */
typedef struct {
    float *gates;
    const float *RT;
    const float *w_f, *w_i, *w_o;
    const float *cs;
    const float *dh;
    float *dpeep;
    float *dbias;
    int reverse;
    uint16_t *dz_bf16;   /* lc_lstm_bwd_bf16 only, may be NULL: [T,B,4N] bf16 (round-to-nearest-even) copy of dz, written in
                          * the same pass (the operand of dX = dz.Kx^T) */
    int shadow_only;     /* lc_lstm_bwd_bf16 with dz_bf16: non-zero = the caller reads dz only through dz_bf16 (every product of
                          * a c5 backward does); `gates` is then UNSPECIFIED after the call.  dpeep / dbias are exact as always. */
} lc_lstm_bwd_dir_t;
size_t lc_lstm_bwd_workspace_bytes(int B, int N, int ndir);
int lc_lstm_bwd(const lc_lstm_bwd_dir_t *dirs /* host array */, int ndir, const int *seq_len, int T, int B,
                int N, void *workspace, size_t workspace_bytes, lc_stream_t stream);

/* bf16-operand variants (BASELINE.json configs[4]): identical contract, except that the two operands of the
 * step GEMM - the recurrent state m'_{t-1} (resp. dz_{t'}) and R (resp. R^T) - are rounded to bf16
 * (round-to-nearest-even) and multiplied by v_mfma_f32_16x16x32_bf16 with float32 accumulation.  Gates, cell
 * state, saved activations and every output stay float32.  Requires num_neurons % 32 == 0. */
int lc_lstm_fwd_bf16(const lc_lstm_fwd_dir_t *dirs /* host array */, int ndir, const int *seq_len, int T, int B,
                     int N, float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream);
int lc_lstm_bwd_bf16(const lc_lstm_bwd_dir_t *dirs /* host array */, int ndir, const int *seq_len, int T, int B,
                     int N, void *workspace, size_t workspace_bytes, lc_stream_t stream);

/* Split-operand variants (config key compute_dtype = bf16x3): identical contract and fp32-grade results - the step product
 * m'_{t-1} . R (BPTT: dz_{t'} . R^T) is computed as in lc_gemm_bf16x3_nt: both fp32 operands split exactly into three bf16
 * terms (the forward state by the consumer, from the same tagged fp32 exchange fragments the fp32 kernels use; dz by its
 * producer; R once per call), the six term pairs of weight >= 2^-16 accumulated in fp32 on v_mfma_f32_16x16x32_bf16 (error
 * bound above).  Gates, cell state, saved activations and every output are the fp32 kernels'.  Split-operand kernels exist
 * for the XCD-pair schedule at num_neurons 768 / 1024 (schedule 6) and for the single-XCD schedule at num_neurons 64 .. 512
 * in steps of 64 (schedule 7), both passes each; the BPTT's exchange carries producer-split bf16 pieces, and its workspace
 * is the larger one lc_lstm_bwd_workspace_bytes already reports.  Every other shape - and the launch-train fall-back of a
 * failed persistent launch - runs the fp32 kernels of lc_lstm_fwd / lc_lstm_bwd (the same arithmetic in another summation order).
 * lc_lstm_bwd_x3 reads dirs[i].dz_bf16 (may be NULL) as the x3 SHADOW of dz - [T * B, 12 N] bf16 in the lc_split_bf16x3 layout,
 * the operand lc_gemm_bf16x3_nt / _tn read for dX / dKx / dR: the split-operand kernels' producers, which split dz for the
 * exchange anyway, write it (bit 18 of the schedule word); every other schedule gets lc_split_bf16x3 behind the recurrence. */
int lc_lstm_fwd_x3(const lc_lstm_fwd_dir_t *dirs /* host array */, int ndir, const int *seq_len, int T, int B,
                   int N, float forget_bias, void *workspace, size_t workspace_bytes, lc_stream_t stream);
int lc_lstm_bwd_x3(const lc_lstm_bwd_dir_t *dirs /* host array */, int ndir, const int *seq_len, int T, int B,
                   int N, void *workspace, size_t workspace_bytes, lc_stream_t stream);

/* DropoutWrapper(output_keep_prob) on a layer output (bilstm.py:128,149):
 *   y[r,p] (+)= x[r,p] * Bernoulli(keep)/keep, the mask being a counter-based hash of
 *   (seed, stream_id, r*P+p) - regenerated, never stored.  x == y (in place) is allowed. */
int lc_dropout_scale(const float *x, int rows, int P, int ldx, float keep, uint32_t seed,
                     uint32_t stream_id, float *y, int ldy, int accumulate, lc_stream_t stream);
/* The same pass also writing the bf16 (round-to-nearest-even) copy of the result, y16 [rows, P] with row pitch ld16:
 * the operand shadow the next bf16 product reads (compute_dtype = bf16), without a separate lc_cast_bf16 pass over the
 * tensor.  P, ldx, ldy, ld16 multiples of 4; x / y 16-byte aligned, y16 8-byte aligned. */
int lc_dropout_scale_bf16(const float *x, int rows, int P, int ldx, float keep, uint32_t seed, uint32_t stream_id,
                          float *y, int ldy, int accumulate, uint16_t *y16, int ld16, lc_stream_t stream);

/* ------------------------------------------------------------------ MoE head ---------------- */
/* create_moe — nnet/moe.py:29-72, fused: logits[r,v] = sum_e softmax_E(a)[r,e] * tau*tanh(q[r,e*V+v]),
 * a = h.Wp+bp [R,E], q = h.W+b [R,E*V] (both produced by lc_gemm_f32). */
/* q is overwritten with tanh(q) (fwd) and then with dq (bwd); pi [R,E] keeps the softmax; da [R,E]. */
int lc_moe_combine_fwd(const float *a, float *q, int R, int E, int V, float tau, float keep,
                       uint32_t seed, float *logits, float *pi, lc_stream_t stream);
int lc_moe_combine_bwd(const float *pi, float *q, const float *dlogits, int R, int E, int V,
                       float tau, float keep, uint32_t seed, float *da, lc_stream_t stream);

/* ------------------------------------------------------------------ optimizer --------------- */
/* L2 (1e-5 * theta on the first n_decay elements) + global-norm clip + optimizer update over ONE flat
 * parameter buffer — nnet/graph.py:183-200.  optimizer: 0 sgd, 1 momentum(0.9),
 * 2 adam(0.9,0.999,1e-8, TF epsilon placement).  state: [n] (momentum) or [2n] (adam m|v).
 * norm_out: device float[2] = {global norm, clip scale}. */
/* guard: NULL, or a device int; when *guard != 0 at execution time the call leaves params and state untouched
 * (norm_out is still written) - see LC_LSTM_STATUS_OFFSET. */
int lc_optimizer_step(float *params, float *grads, size_t n, size_t n_decay, float l2,
                      float clip_norm, int optimizer, float lr, int step, float *state,
                      float *norm_out, const int *guard, void *workspace, size_t workspace_bytes,
                      lc_stream_t stream);
size_t lc_optimizer_workspace_bytes(size_t n);

/* column sums: out[N] (+)= sum_rows x[rows,N]  (bias gradients).  Deterministic: row slabs are summed into the
 * workspace (lc_colsum_workspace_bytes) and folded in a fixed order.  The workspace is REQUIRED: at least
 * N * sizeof(float) bytes (LC_EINVAL otherwise); one smaller than lc_colsum_workspace_bytes(N) selects a single slab
 * (same result for the same call, slower). */
size_t lc_colsum_workspace_bytes(int N);
int lc_colsum(const float *x, int rows, int N, int ldx, float *out, int accumulate, void *workspace,
              size_t workspace_bytes, lc_stream_t stream);
/* out[cols,rows] = in[rows,cols]^T */
int lc_transpose(const float *in, int rows, int cols, float *out, lc_stream_t stream);

/* KL label-smoothing regulariser of nnet/bilstm.py:255-269 over ALL rows (padded frames included):
 *   *loss_acc += weight * sum p*(log p - log q)   (device double, caller zeroes it);  q = uniform if log_q NULL
 *   dlogits  += its gradient (may be NULL). */
int lc_label_smoothing(const float *logits, int rows, int V, const float *log_q, float weight,
                       double *loss_acc, float *dlogits, lc_stream_t stream);

/* Softmax posteriors for nnet-forward: out = softmax(smooth*logits) or its log, minus prior
 * (nnet/graph.py:236, bin/nnet-forward.py:87-91). prior may be NULL. */
int lc_posteriors(const float *logits, int rows, int V, float smooth, int apply_softmax,
                  int apply_log, const float *log_prior, float *out, lc_stream_t stream);

/* ------------------------------------------------------------------ batch normalisation ----- */
/* tf.layers.batch_normalization as create_logits_lstm uses it (nnet/lstm.py:271-294; rank-3 input => TF's
 * non-fused path): per-column moments over ALL rows of x[rows, C] (rows = T*B, padded frames included),
 * population variance; y = (x - mean) * rsqrt(var + eps) * gamma + beta (eps = 1e-3 in the reference).
 *   lc_bn_moments        batch mean / variance (training)            workspace: lc_bn_workspace_bytes(C)
 *   lc_bn_apply          normalise with the given mean / var (batch moments, or the moving averages at inference)
 *   lc_bn_bwd            dx, dgamma, dbeta (overwritten); training != 0: gradient through the batch moments
 *   lc_bn_update_moving  assign_moving_average, v -= (v - batch) * (1 - momentum), the UPDATE_OPS of
 *                        nnet/graph.py:194-196 (momentum = 0.99)
 * dx may alias dy. */
size_t lc_bn_workspace_bytes(int C);
int lc_bn_moments(const float *x, int rows, int C, int ldx, float *mean, float *var, void *workspace,
                  size_t workspace_bytes, lc_stream_t stream);
int lc_bn_apply(const float *x, int rows, int C, int ldx, const float *mean, const float *var,
                const float *gamma, const float *beta, float eps, float *y, int ldy, lc_stream_t stream);
int lc_bn_bwd(const float *x, const float *dy, int rows, int C, int ldx, int lddy, const float *mean,
              const float *var, const float *gamma, float eps, int training, float *dx, int lddx,
              float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, lc_stream_t stream);
int lc_bn_update_moving(float *moving_mean, float *moving_var, const float *mean, const float *var, int C,
                        float momentum, lc_stream_t stream);
/* dynamic_rnn's zero output beyond sequence_length, re-applied to a time-major x[T*B, C] whose padded rows are no
 * longer zero (a ResidualWrapper input that went through batch normalisation): x[t*B+b, :] = 0 for t >= seq_len[b]. */
int lc_length_mask(float *x, int T, int B, int C, int ldx, const int *seq_len, lc_stream_t stream);

/* ------------------------------------------------------------------ development hook -------- */
/* Not part of the product surface: when set to a device buffer of [T][4 waves][8] 64-bit words, one workgroup of
 * the forward step kernel stores s_memtime stamps of its phases there (tools/stamp_probe.py); NULL switches it off. */
void lc_debug_set_lstm_stamps(unsigned long long *buf);
/* Which schedule the calling thread's last lc_lstm_fwd* / lc_lstm_bwd* call took (tests assert it):
 *   bits 0-7   1 = persistent float32, 2 = persistent bf16, 3 = two-stream launch train, 4 = launch train,
 *              5 = persistent float32 over XCD pairs (num_neurons 640 / 768 / 896 / 1024),
 *              6 = split-operand (bf16x3) recurrence over XCD pairs (num_neurons 768 / 1024), 7 = split-operand, one XCD
 *              per (direction, row group) (num_neurons 64 .. 512 in steps of 64)
 *   bits 8-15  row tiles of 16 per workgroup (launch train), bit 16 = bf16 operands, bit 17 = backward,
 *   bit 18 = the x3 shadow of dz was written by the BPTT kernel itself (lc_lstm_bwd_x3). */
int lc_debug_last_lstm_schedule(void);
/* Same kind of hook for the CTC scan: device buffer of [2 phases][5 waves][512 iterations][8] 64-bit s_memtime stamps
 * of workgroup 0 (tools/ctc_stamps.py); NULL switches it off. */
void lc_debug_set_ctc_stamps(unsigned long long *buf);
/* A FOREIGN resident kernel for tests: `blocks` workgroups of 256 threads that stay resident for `microseconds` of wall
 * clock, hold `lds_bytes` of LDS each and do nothing else - what a collective's kernel waiting for a slower peer looks like
 * to the next persistent recurrence (one workgroup with >= 84 KB of LDS per CU on every CU of an XCD: with lds_bytes >= 80 KB
 * the two cannot share a CU).  tests/test_gpu_coresidency.py rehearses that hazard with it. */
int lc_debug_spin(int blocks, int microseconds, int lds_bytes, lc_stream_t stream);
/* The whole-round rule of the 256 x 256 product kernels (HOST arithmetic, no GPU needed): of `row_tiles` x `col_tiles` tiles
 * of an unsplit product, how many row tiles the big kernel takes so that its rounds of `cus` workgroups (0 = the current
 * device's CU count, 256 without a device) are whole; the rows behind them run on the 128 x 128 kernel.  tests/test_host.py. */
int lc_debug_gemm_whole_round_row_tiles(int row_tiles, int col_tiles, int cus);

/* ------------------------------------------------------------------ input path (HOST) ------- */
/* TFRecord + tf.train.SequenceExample decoding without TensorFlow: what tf.data.TFRecordDataset(...).map(_parse,
 * num_parallel_calls) does per utterance in nnet/tfrecord.py:94-125, with _splice (28-40) and _subsample (43-51) fused
 * into the copy.  HOST functions on HOST buffers, no GPU involved, thread-safe and re-entrant: the loader calls them from
 * `--num-parallel-calls` worker threads (ctypes releases the GIL).  `file` is the whole .tfrecords file image; its FIRST
 * record is the utterance (the reference's converter writes one SequenceExample per file, tfrecord.py:128-156).
 *   lc_tfrecord_inspect: validates the framing - with verify_crc != 0 both masked CRC-32C words, as TF's reader does
 *     (a mismatch is LC_EINVAL "corrupted record") - and counts: frames and dim of feature list "nnet_input" (every frame
 *     must have the same number of floats), steps of "nnet_target".
 *   lc_tfrecord_decode: writes row j of the spliced / subsampled matrix to x + j * x_row_stride (floats), j < T',
 *     T' = subsample ? T / subsample : T, row width dim * (1 + left + right), edge frames replicated; and the labels
 *     (one int64 per step) to labels[].  x or labels may be NULL.  A row stride larger than the row width lets the caller
 *     decode straight into one utterance's rows of a padded TIME-MAJOR batch [T, B, D'] (stride B * D').
 *   lc_crc32c: plain (unmasked) CRC-32C (Castagnoli), hardware instruction when the CPU has it. */
typedef struct {
    int64_t num_frames;  /* T: features in "nnet_input" (before subsampling) */
    int32_t dim;         /* floats per frame (before splicing) */
    int32_t has_input, has_target;
    int64_t num_labels;  /* features in "nnet_target" */
} lc_seqex_info_t;
int lc_tfrecord_inspect(const void *file, size_t nbytes, int verify_crc, lc_seqex_info_t *info);
int lc_tfrecord_decode(const void *file, size_t nbytes, int dim, int left_context, int right_context, int subsample,
                       float *x, size_t x_row_stride, int64_t max_rows, int64_t *labels, int64_t max_labels);
uint32_t lc_crc32c(const void *data, size_t nbytes);
/* A whole padded batch in two calls, each fanned out over `nthreads` native threads (the tf.data map + padded_batch of
 * tfrecord.py:122-123 / pipeline.py:35-61).  lc_batch_open reads the n files, verifies their CRCs and reports raw frame
 * and label counts (num_labels[i] = -1 for a record without an "nnet_target" list: a caller whose tfrecords.scp says
 * has_label = 1 must treat that as an error, as tf.parse_single_sequence_example does for a missing feature list, tfrecord.py:
 * 94-105; dimension checked against expect_dim when > 0); a file with bytes behind its first record is refused; the caller sizes the batch; lc_batch_decode writes
 * utterance i's row j to x + i * utt_stride + j * row_stride (floats) - time-major [T,B,D']: utt_stride = D',
 * row_stride = B * D'; batch-major [B,T,D']: utt_stride = T * D', row_stride = D' - zero-fills its rows up to max_rows,
 * and writes its labels to labels + i * label_stride, padded with pad_label up to max_labels.  The first failing file
 * ends the call with its path in lc_last_error().  lc_batch_close frees the reader (always call it after a
 * successful open). */
typedef struct lc_batch_reader lc_batch_reader_t;
int lc_batch_open(const char *const *paths, int n, int verify_crc, int expect_dim, int nthreads,
                  lc_batch_reader_t **reader, int64_t *num_frames, int64_t *num_labels);
int lc_batch_decode(lc_batch_reader_t *reader, int left_context, int right_context, int subsample, float *x,
                    size_t utt_stride, size_t row_stride, int64_t max_rows, int64_t *labels, size_t label_stride,
                    int64_t max_labels, int64_t pad_label, int nthreads);
void lc_batch_close(lc_batch_reader_t *reader);

#ifdef __cplusplus
}
#endif
#endif
