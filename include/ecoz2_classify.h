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

#ifdef __cplusplus
}
#endif
#endif
