/*
 * ecoz2_classify.h -- C-ABI of the consumers of the VQ path's output in libecoz2vq.so:
 * the symbol-sequence classifiers that read the `.seq` files `ecoz2 vq quantize` writes, and the HMM path that
 * quantises `.prd` files on the fly (`hmm classify --predictors --codebooks`).
 *
 * Part A (nb / mm / c12n) restates host code that IS present in the reference as Rust; each entry point cites the
 * function it mirrors.  These are host-side (CPU) in the reference and stay host-side here: they are O(symbols) table
 * look-ups over the GPU path's output, with the reference's exact arithmetic (f64 for nb, f32 for mm, libm log10).
 * Part B (hmm) replaces the reference's FFI symbols for the HMM commands (src/ecoz2_lib/mod.rs:134-167); the arithmetic
 * behind them lives in the absent C submodule, so its definitions are this repo's; scoring and training
 * run in HIP kernels, with no CPU fallback.
 *
 * Plain pointers and sizes only.  Every function returns 0 on success; on failure a message is on stderr and in
 * e2vq_last_error().
 */
#ifndef ECOZ2_CLASSIFY_H
#define ECOZ2_CLASSIFY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- .seq reader: sequence::load, src/sequence/mod.rs:49-75 ------------------------------------------------- */
int e2vq_seq_info(const char *path, char class_name[96], int *M, int64_t *T);
int e2vq_seq_read(const char *path, uint16_t *sym, int64_t capacity);

/* =========================================================================================
 * Part A -- nb / mm / c12n (pure Rust in the reference)
 * ======================================================================================= */

/* nbayes::learn (src/nb/nbayes.rs:63-114) + main_nbayes_learn (src/nb/mod.rs:99-128): symbol frequencies of the
 * given sequences -> CBOR model `<out_root>/data/nbs/M<M>/<class>.nb` (utl::save_ser, src/utl/mod.rs:263-268).
 * out_path (optional, may be NULL) receives the file name written. */
int ecoz2_nb_learn(int codebook_size, const char *const *seq_filenames, int num_sequences, char *out_path,
                   int out_path_cap);
/* nbayes::classify (src/nb/nbayes.rs:116-153): log10-probability of every sequence under every model (m-estimate,
 * f64, summed in symbol order), c12n report on stdout, `nb_<M>_classification.json` / `nb_<M>_y_true_pred.json`. */
int ecoz2_nb_classify(const char *const *nb_filenames, int num_models, const char *const *seq_filenames,
                      int num_sequences, int show_ranked, int codebook_size);
/* NBayes::show (src/nb/nbayes.rs:21-35) */
int ecoz2_nb_show(const char *nb_filename);
/* NBayes::log_prob_sequence (src/nb/nbayes.rs:50-54) of one sequence file under one model file */
int e2vq_nb_log_prob(const char *nb_filename, const char *seq_filename, double *log_prob);

/* markov::learn (src/mm/markov.rs:59-126) + main_mm_learn (src/mm/mod.rs:99-128): first-order Markov model of the
 * symbol sequences (f32 counters with add-one smoothing) -> CBOR `<out_root>/data/mms/M<M>/<class>.mm`.
 * The row-stochastic asserts of markov.rs:117,122 are checked (failure = error). */
int ecoz2_mm_learn(int codebook_size, const char *const *seq_filenames, int num_sequences, char *out_path,
                   int out_path_cap);
/* markov::classify (src/mm/markov.rs:128-167) */
int ecoz2_mm_classify(const char *const *mm_filenames, int num_models, const char *const *seq_filenames,
                      int num_sequences, int show_ranked, int codebook_size);
/* MM::show (src/mm/markov.rs:27-40), asserts included */
int ecoz2_mm_show(const char *mm_filename);
/* MM::log_prob_sequence (src/mm/markov.rs:43-49): f32 */
int e2vq_mm_log_prob(const char *mm_filename, const char *seq_filename, float *log_prob);

/* C12nResults (src/c12n/mod.rs:8-223) driven directly: `probs` is num_cases x num_models (row-major), `class_ids`
 * the true model index of every case.  Prints exactly what add_case / report_results print and writes the two JSON
 * files `<out_base_name>_classification.json`, `<out_base_name>_y_true_pred.json`.  result / confusion (optional)
 * receive the (num_models+1)^2 tables. */
int e2vq_c12n_run(const char *const *model_class_names, int num_models, const int *class_ids,
                  const char *const *case_class_names, const char *const *case_titles, const double *probs,
                  int num_cases, int show_ranked, const char *out_base_name, int *result, int *confusion);

/* =========================================================================================
 * Part B -- hmm (the reference's FFI symbols; bodies in the absent C submodule -> definitions in DESIGN.md /
 *           oracle/hmm_oracle.h; parity unpinned)
 * ======================================================================================= */

/* replaces `fn ecoz2_set_random_seed(seed: c_long) -> c_ulong`          src/ecoz2_lib/mod.rs:75
 * seed < 0: time based (src/hmm/mod.rs:73-76).  Returns the seed in use.  Feeds the random model types of hmm learn. */
unsigned long ecoz2_set_random_seed(long seed);

/* `callback: extern "C" fn(*mut c_char, c_double)`                       src/ecoz2_lib/mod.rs:144
 * called once per E-step with ("sum_log_prob", sum over the training sequences of ln P(O | model)) */
typedef void (*ecoz2_hmm_learn_callback_t)(char *variable, double value);

/* replaces `fn ecoz2_hmm_learn(N, model_type, sequence_filenames, num_sequences: c_uint, hmm_epsilon, val_auto,
 *           max_iterations, use_par, callback)`                          src/ecoz2_lib/mod.rs:134-145
 * Baum-Welch over all the given `.seq` files of ONE class (class name and M from the first sequence): E-step on the
 * GPU (one wavefront per sequence, exact fixed-point expected counts), M-step on the GPU, until the increase of
 * sum ln P drops to val_auto or max_iterations (>= 0) E+M steps ran.  hmm_epsilon > 0 floors B and renormalises.
 * Writes data/hmms/N<N>__M<M>_t<type>__a<val_auto>[_I<max>]/<class>.hmm (+ .csv with the measure per iteration).
 * use_par is accepted and ignored.
 * Limits: 1 <= N <= 512 (up to 64 states one lane of a wavefront per state, beyond a workgroup per sequence with a thread
 * per state -- same arithmetic, slower; the reference's -N is free: larger N fail with a message).
 * An empty sequence is skipped by the training (no counts, not in the pi denominator) and scores P = 1 in classify.
 * ECOZ2_VQ_GPUS = W: the sequences are dealt to W workers (devices ECOZ2_VQ_DEVICE + w modulo the device count); the
 * exact int64 expected counts are summed over the workers each E-step, so the model is the single worker's bit for
 * bit; ecoz2_hmm_classify / ecoz2_hmm_classify_predictors deal the sequences / predictor files the same way. */
int ecoz2_hmm_learn(int N, int model_type, const char *const *sequence_filenames, unsigned num_sequences,
                    double hmm_epsilon, double val_auto, int max_iterations, int use_par,
                    ecoz2_hmm_learn_callback_t callback);

/* replaces `fn ecoz2_hmm_classify(model_filenames, num_models: c_uint, sequence_filenames, num_sequences: c_uint,
 *           show_ranked, classification_filename)`                       src/ecoz2_lib/mod.rs:147-154
 * ln P(O | model) of every sequence under every model (GPU, one wavefront per pair); report in the layout of
 * src/c12n/mod.rs; classification_filename (may be NULL): the CSV of CHANGELOG.md:273-284. */
int ecoz2_hmm_classify(const char *const *model_filenames, unsigned num_models,
                       const char *const *sequence_filenames, unsigned num_sequences, int show_ranked,
                       const char *classification_filename);

/* replaces `fn ecoz2_hmm_classify_predictors(model_filenames, num_models: c_uint, cb_filenames, num_codebooks: c_int,
 *           prd_filenames, num_predictors: c_int, show_ranked, classification_filename)`   src/ecoz2_lib/mod.rs:156-165
 * The on-the-fly consumer of the nearest-codeword kernel: every `.prd` is uploaded once, quantised on the GPU against
 * the codebook of each model (a single codebook serves all models; several are matched to the models by class name)
 * and scored where the symbols are. */
int ecoz2_hmm_classify_predictors(const char *const *model_filenames, unsigned num_models,
                                  const char *const *cb_filenames, int num_codebooks,
                                  const char *const *prd_filenames, int num_predictors, int show_ranked,
                                  const char *classification_filename);

/* replaces `fn ecoz2_hmm_show(hmm_filename, format)`                      src/ecoz2_lib/mod.rs:167
 * format: one printf floating-point conversion per value, default "%Lg " (src/hmm/mod.rs:153-154) */
int ecoz2_hmm_show(const char *hmm_filename, const char *format);

/* ---- array-level entry points over the same kernels (tests, Python mirror) ---------------------------------- */
/* initial model of `hmm learn -t`: 0 random, 1 uniform, 2 cascade-2, 3 cascade-3 (random B); uses the generator
 * seeded by ecoz2_set_random_seed.  pi[N], A[N*N], B[N*M]; N <= 512. */
int e2vq_hmm_init(int N, int M, int model_type, double *pi, double *A, double *B);
int e2vq_hmm_save(const char *path, const char *class_name, int N, int M, const double *pi, const double *A,
                  const double *B);
int e2vq_hmm_info(const char *path, char class_name[96], int *N, int *M);
int e2vq_hmm_load(const char *path, double *pi, double *A, double *B);
/* scaled forward pass of S sequences (concatenated u16 symbols + S+1 offsets) under K models sharing M; outputs at
 * [s*K + k]: P(O) = mant * 2^exp2 (mant in [0.5,1)), status (0 ok, 1 cannot emit, 2 symbol >= M), ln P */
int e2vq_hmm_score(int device, int K, const int *Ns, int M, const double *const *pis, const double *const *As,
                   const double *const *Bs, const uint16_t *sym, const int64_t *offs, int S, double *mant,
                   int64_t *exp2, int *status, double *log_probs);
/* int64 words of the E-step accumulators: [hi, lo] limb pairs  PI[N] | AN[N][N] | AD[N] | BN[N][M] | BD[N] | used, skipped */
int64_t e2vq_hmm_acc_words(int N, int M);
int e2vq_hmm_estep(int device, int N, int M, const double *pi, const double *A, const double *B, const uint16_t *sym,
                   const int64_t *offs, int S, int64_t *acc, double *mant, int64_t *exp2, int *status);
/* the loop of ecoz2_hmm_learn on arrays, in place; sum_log_prob[0..*num_esteps) = the measure per E-step */
int e2vq_hmm_train(int device, int N, int M, double *pi, double *A, double *B, const uint16_t *sym,
                   const int64_t *offs, int S, double epsilon, double val_auto, int max_iterations,
                   double *sum_log_prob, int cap, int *num_esteps);

#ifdef __cplusplus
}
#endif
#endif
