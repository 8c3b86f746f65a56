/*
 * hmm_oracle.h -- CPU ORACLE for the HMM consumers of the VQ path:
 * `hmm learn`, `hmm classify` (sequences / predictors+codebooks), `hmm show`.
 *
 * TEST INFRASTRUCTURE ONLY (tests/ and __graft_entry__.smoke() may load it, as the checker).
 *
 * PARITY UNPINNED.  The reference binds these commands to C functions (ecoz2_hmm_learn, ecoz2_hmm_classify,
 * ecoz2_hmm_classify_predictors, ecoz2_hmm_show, ecoz2_set_random_seed: src/ecoz2_lib/mod.rs:75,134-167) whose
 * bodies (ecoz2/src/hmm/{hmm,hmm_learn,hmm_classify,hmm_show,hmm_adjustb,hmm_file,hmm_prob,hmm_log_prob,
 * hmm_genQopt,hmm_estimateB,hmm_gen,distr,symbol}.c, build.rs:34-48) are absent from /root/reference (empty
 * submodule); no fixture or test of theirs exists.  What reference text pins and this oracle follows:
 *   - the FFI signatures and the CLI options with their defaults (src/hmm/mod.rs:40-145: N=5, type 3,
 *     I=-1, epsilon=1e-05 "epsilon restriction on B, 0 = do not apply", val_auto=0.3, seed, model types
 *     0 random / 1 uniform / 2 cascade-2 random B / 3 cascade-3 random B);
 *   - prob_t = double (CHANGELOG.md:178); model directory naming data/hmms/N<N>__M<M>_t<type>__a<val_auto>_I<I>
 *     (CHANGELOG.md:460); the classification CSV of `--c12n` (CHANGELOG.md:273-284); the report layout, which
 *     src/c12n/mod.rs:5 says is a translation of the C report.
 * Everything else below (file layout, generator, scaled Baum-Welch with exact fixed-point sums, the stopping
 * rule) is this repo's own strict-IEEE definition, normative for the HIP implementation.
 */
#ifndef HMM_ORACLE_H
#define HMM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E2H_MAX_N 512
#define E2H_MAX_ESTEPS 1000 /* safety cap (val_auto <= 0 with no iteration limit would never stop) */
#define E2H_ACC_SHIFT 29 /* accumulated quantities are < 2: x ~= (hi*2^31 + lo) * 2^-(29+31) */

/* ---- generator (ecoz2_set_random_seed) ----------------------------------------------------- */
/* seed < 0: time based.  Returns the seed in use.  splitmix64 stream; uniform = (x >> 11) * 2^-53 */
uint64_t e2h_set_random_seed(int64_t seed);
double e2h_uniform(void);

/* ---- model ------------------------------------------------------------------------------------ */
/* pi[N], A[N*N] (row i = from-state), B[N*M] (row j = state).  type: 0 random pi/A/B, 1 uniform,
 * 2 cascade-2 (i -> i, i+1), 3 cascade-3 (i -> i, i+1, i+2), pi = e_0, random B.  Random rows are uniform
 * draws divided by their sequential sum.  Draw order: pi, A (row-major), B (row-major). */
int e2h_init(int N, int M, int type, double *pi, double *A, double *B);

/* ---- scoring ---------------------------------------------------------------------------------- */
/* Scaled forward pass.  alpha~_0(i) = pi_i * B_i(o_0); alpha~_t(j) = (chain_i fma(alpha^_{t-1}(i), A_ij, acc)) * B_j(o_t);
 * c_t = sequential sum_j alpha~_t(j); alpha^_t = alpha~_t / c_t.  P(O) = prod c_t is returned as mant * 2^exp2
 * with mant in [0.5, 1): per step (m, e) = frexp(c_t); p = p * m; (p, e') = frexp(p); E += e + e'.
 * Returns 0, or 1 if some c_t == 0 (the model cannot emit the sequence; mant = 0, exp2 = 0). T = 0: P = 1. */
int e2h_forward(int N, int M, const double *pi, const double *A, const double *B, const uint16_t *o, int64_t T,
                double *mant, int64_t *exp2, double *alpha_hat /* T*N or NULL */, double *c /* T or NULL */);
/* natural log of mant * 2^exp2: log(mant) + (double)exp2 * M_LN2 (libm); -INFINITY when mant == 0 */
double e2h_log_prob(double mant, int64_t exp2);

/* ---- training ------------------------------------------------------------------------------- */
/* accumulator words (int64): [hi, lo] pairs: PI[N] | AN[N][N] | AD[N] | BN[N][M] | BD[N] | used, skipped */
int64_t e2h_acc_words(int N, int M);
/* forward-backward on one sequence, adding its expected counts to acc (exact fixed point):
 *   u_j = (B_j(o_{t+1}) * beta^_{t+1}(j)) / c_{t+1};  beta^_t(i) = chain_j fma(A_ij, u_j, acc);  beta^_{T-1} = 1
 *   gamma_t(i) = alpha^_t(i) * beta^_t(i);  xi_t(i,j) = (alpha^_t(i) * A_ij) * u_j
 *   PI += gamma_0;  AN += xi_t, AD += gamma_t (t < T-1);  BN[.][o_t] += gamma_t, BD += gamma_t (all t)
 * Sequences the model cannot emit (or empty ones) are skipped and counted in acc[...skipped]. */
int e2h_accumulate(int N, int M, const double *pi, const double *A, const double *B, const uint16_t *o, int64_t T,
                   int64_t *acc, double *mant, int64_t *exp2);
/* pi_i = PI_i / used;  A_ij = AN_ij / AD_i (AD_i > 0);  B_jk = BN_jk / BD_j (BD_j > 0); then, if epsilon > 0,
 * every B_jk < epsilon becomes epsilon and the row is divided by its sequential sum (hmm_adjustb) */
void e2h_reestimate(int N, int M, const int64_t *acc, double epsilon, double *pi, double *A, double *B);

typedef void (*e2h_learn_callback_t)(char *variable, double value);
/* Whole training: it = 0; loop { if ((max_iterations >= 0 && it >= max_iterations) || it >= E2H_MAX_ESTEPS) stop; E-step over all sequences
 * (L_it = sequential sum of log P of the used sequences); callback("sum_log_prob", L_it);
 * if (it > 0 && L_it - L_{it-1} <= val_auto) stop (model of this E-step kept); M-step; it++ }.
 * sum_log_prob receives the L_it (capacity cap); returns the number of E-steps run, < 0 on error. */
int e2h_learn(int N, int M, const uint16_t *const *seqs, const int64_t *lens, int R, double epsilon, double val_auto,
              int max_iterations, double *pi, double *A, double *B, double *sum_log_prob, int cap,
              e2h_learn_callback_t callback);

/* ---- .hmm files: 16-byte ident "<hmm>", 96-byte class name (src/utl/mod.rs:19-20), u32 N, u32 M,
 *      then pi[N], A[N*N], B[N*M] as little-endian f64 ------------------------------------------ */
int e2h_save(const char *path, const char *class_name, int N, int M, const double *pi, const double *A, const double *B);
int e2h_load_info(const char *path, char class_name[96], int *N, int *M);
int e2h_load(const char *path, double *pi, double *A, double *B);

#ifdef __cplusplus
}
#endif
#endif
