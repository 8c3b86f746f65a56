/*
 * vq_oracle.h -- CPU ORACLE for the ecoz2 VQ hot path (vq learn / vq quantize).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the C library
 * github.com/ecoz2/ecoz2 (git submodule `ecoz2` of the reference, .gitmodules:1-3;
 * file list in build.rs:10-49), which is absent from /root/reference and whose
 * pinned commit is unrecoverable.  The reference holds no test, golden vector or
 * fixture for vq learn / vq quantize.  What IS pinned by reference source text and
 * checked in tests/: the .seq byte layout (src/sequence/mod.rs:49-75), the header
 * constants (src/utl/mod.rs:19-20), the Levinson recursion (src/lpc/lpca_r_rs.rs:8-43,
 * src/lpc/lpca_rs.rs:28-75 with fixture signal_frame.inputs), the FFI signatures
 * (src/ecoz2_lib/mod.rs:96-122) and the shape of the LBG loop visible in the run log
 * (notes.md:122-153).  Everything else is this repo's own strict-IEEE definition,
 * written down in DESIGN.md and below; it is normative for the HIP implementation.
 */
#ifndef VQ_ORACLE_H
#define VQ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E2O_FILE_IDENT_LEN 16      /* src/utl/mod.rs:19 */
#define E2O_MAX_CLASS_NAME_LEN 96  /* src/utl/mod.rs:20 */
#define E2O_MAX_P 200              /* CHANGELOG.md:183 "increased maximum prediction order (200)" */

/* ---- LPC pieces ------------------------------------------------------- */

/* Autocorrelation + Levinson-Durbin on a signal frame; src/lpc/lpca_rs.rs:28-75 (lpca1).
 * r, rc, a have P+1 entries. Returns 0 ok, 1 if r[0]==0, 2 if the error went <= 0. */
int e2o_lpca(const double *x, int n, int P, double *r, double *rc, double *a, double *pe);

/* Levinson-Durbin from a given autocorrelation; src/lpc/lpca_r_rs.rs:8-43 (lpca_r). */
int e2o_lpca_r(int P, const double *r, double *rc, double *a, double *pe);

/* reflections rc[1..P] -> predictor a (step-up, same ops as lpca_r's inner update)
 * -> raa[n] = sum_{i=0}^{P-n} a[i]*a[i+n]  (build.rs:23 ref2raas.c, absent). */
void e2o_ref2raas(int P, const double *rc, double *raa);

/* cq[0] = raa[0]; cq[n] = 2*raa[n]: the pre-doubled codeword the sweep multiplies with. */
void e2o_codeword_q(int P, const double *raa, double *cq);

/* d(r, c) = r0*c0 + 2*sum r[n]*c[n] evaluated as the canonical chain
 *   acc = +0.0; for n = 0..P: acc = fma(r[n], cq[n], acc)               
 * (one rounding per term, ascending n; exactly what v_mfma_f64_16x16x4_f64 computes per k) */
double e2o_distortion(int P, const double *r, const double *cq);

/* ---- fixed-point exact accumulation (order-free sums) ------------------ */

/* x -> two signed limbs, x ~= (hi*2^31 + lo) * 2^-(sh+31), |hi|,|lo| <= 2^30 */
void e2o_fix(double x, int sh, int64_t *hi, int64_t *lo);
/* (sum_hi*2^31 + sum_lo) correctly rounded to double, times 2^-(sh+31) */
double e2o_unfix(int64_t sum_hi, int64_t sum_lo, int sh);

/* scale exponents */
int e2o_shift_frames(double maxabs);              /* sh_r  = 29 - ilogb(maxabs)   */
int e2o_shift_frames_sq(double maxabs);           /* sh_q  = 28 - 2*ilogb(maxabs) */
/* Ed from the codebook: L = max_m sum_n |cq_m[n]|, B = maxabs*L + 1, Ed = ilogb(B)+2 */
int e2o_dist_exponent(int P, const double *cq, int M, double maxabs);
/* sh_d = 30 - Ed, sh_d2 = 30 - 2*Ed */

/* accumulator row layout (int64 per cell m): [2*n+limb] cell sums n=0..P, then
 * count, dist_hi, dist_lo, dist2_hi, dist2_lo; stride = roundup(2*(P+1)+5, 8) */
int e2o_row_stride(int P);

/* ---- passes ------------------------------------------------------------ */

/* nearest-codeword assignment only (vq quantize; build.rs:25,33) */
void e2o_quantize(int P, const double *cq, int M, const double *frames, int64_t T,
                  uint16_t *sym, double *dmin);

/* one LBG pass: assignment + exact accumulation into rows[M*stride] (zeroed here).
 * sym/dmin may be NULL. */
void e2o_pass(int P, const double *cq, int M, const double *frames, int64_t T,
              int sh_r, int Ed, uint16_t *sym, double *dmin, int64_t *rows);

/* data statistics computed once per training set */
typedef struct {
    double maxabs;
    int64_t sum_hi[E2O_MAX_P + 1], sum_lo[E2O_MAX_P + 1]; /* global cell (M=1) limbs */
    int64_t q_hi, q_lo;                                    /* sum of squares limbs    */
} e2o_stats;
int e2o_data_stats(int P, const double *frames, int64_t T, e2o_stats *st);

typedef struct {
    double DD, avg, sigma, inertia;
    int64_t empty_cells, failed_cells;
} e2o_level_stats;

/* per-level statistics from reduced rows (does not modify the codebook) */
void e2o_rows_stats(int P, int M, const int64_t *rows, int64_t T, int sh_r, int Ed,
                    double Q, e2o_level_stats *out);
/* centroid update from reduced rows: reflections[M*(P+1)] updated in place for
 * non-empty cells whose Levinson succeeds */
void e2o_update(int P, int M, const int64_t *rows, int sh_r, double *reflections,
                e2o_level_stats *out);
/* M -> 2M split: new[2i] = old[i]*0.99, new[2i+1] = old[i]*1.01 (n = 1..P) */
void e2o_grow(int P, int M, const double *reflections, double *grown);
/* reflections[M*(P+1)] -> cq[M*(P+1)] */
void e2o_reflections_to_cq(int P, int M, const double *reflections, double *cq);

/* ---- LBG driver ---------------------------------------------------------- */

typedef void (*e2o_learn_cb)(void *target, int M, double avg_distortion, double sigma,
                             double inertia);
typedef void (*e2o_level_hook)(void *user, int M, int passes, const double *reflections,
                               const e2o_level_stats *st);

/* frames: T x (P+1) row-major.  base_reflections/base_M may be NULL/0.
 * out_root: directory prefix for data/codebooks/<class>/... (NULL = no files).
 * Returns 0 on success. */
int e2o_learn(int P, double eps, const char *class_name, const double *frames, int64_t T,
              const double *base_reflections, int base_M, int max_M, const char *out_root,
              void *target, e2o_learn_cb cb, void *hook_user, e2o_level_hook hook);

/* ---- files --------------------------------------------------------------- */

int e2o_prd_save(const char *path, const char *class_name, int P, const double *frames, int64_t T);
/* caller frees *frames with e2o_free */
int e2o_prd_load(const char *path, char class_name[E2O_MAX_CLASS_NAME_LEN], int *P,
                 double **frames, int64_t *T);
int e2o_cbook_save(const char *path, const char *class_name, int P, int M, const double *reflections);
int e2o_cbook_load(const char *path, char class_name[E2O_MAX_CLASS_NAME_LEN], int *P, int *M,
                   double **reflections);
int e2o_seq_save(const char *path, const char *class_name, int M, const uint16_t *sym, int64_t T);
int e2o_seq_load(const char *path, char class_name[E2O_MAX_CLASS_NAME_LEN], int *M,
                 uint16_t **sym, int64_t *T);
void e2o_free(void *p);

void e2o_set_threads(int n); /* OpenMP threads for the passes (bench.py cpu_baseline) */
/* timing helper for bench.py's cpu_baseline: runs `reps` passes, returns seconds */
double e2o_time_pass(int P, const double *cq, int M, const double *frames, int64_t T, int reps,
                     int *threads_used);

#ifdef __cplusplus
}
#endif
#endif
