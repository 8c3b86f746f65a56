/*
 * vq_oracle.c -- CPU ORACLE for the ecoz2 VQ hot path.  TEST INFRASTRUCTURE ONLY
 * (see vq_oracle.h).  PARITY UNPINNED by the reference's own tests; the pieces that
 * reference source text does pin are cited at each function.
 *
 * Build: strict IEEE (-O2 -fno-fast-math -ffp-contract=off -mfma); every FMA is an
 * explicit fma().  The "reference-flags" timing variant (-O3 -ffast-math -fopenmp,
 * build.rs:4-6) is compiled from this same file with -DE2O_FAST and is only used as
 * bench.py's cpu_baseline, never as a checker.
 */
#define _GNU_SOURCE
#include "vq_oracle.h"

#include <errno.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __int128 i128;
typedef unsigned __int128 u128;

#ifdef E2O_FAST
/* reference-flags variant: let the compiler contract/reassociate as build.rs:4-6 does */
#define E2O_FMA(a, b, c) ((a) * (b) + (c))
#else
#define E2O_FMA(a, b, c) fma((a), (b), (c))
#endif

/* ------------------------------------------------------------------------ */
/* LPC                                                                      */
/* ------------------------------------------------------------------------ */

/* src/lpc/lpca_r_rs.rs:8-43 */
int e2o_lpca_r(int P, const double *r, double *rc, double *a, double *pe_out)
{
    double pe = 0.;
    const double r0 = r[0];
    if (0.0 == r0) {
        *pe_out = pe;
        return 1;
    }
    pe = r0;
    a[0] = 1.0;
    for (int k = 1; k <= P; k++) {
        double sum = 0.0;
        for (int i = 1; i <= k; i++) sum -= a[k - i] * r[i];
        const double akk = sum / pe;
        rc[k] = akk;
        a[k] = akk;
        for (int i = 1; i <= (k >> 1); i++) {
            const double ai = a[i];
            const double aj = a[k - i];
            a[i] = ai + akk * aj;
            a[k - i] = aj + akk * ai;
        }
        pe *= 1.0 - akk * akk;
        if (pe <= 0.0) {
            *pe_out = pe;
            return 2;
        }
    }
    *pe_out = pe;
    return 0;
}

/* src/lpc/lpca_rs.rs:28-75 (lpca1): autocorrelation then the same recursion */
int e2o_lpca(const double *x, int n, int P, double *r, double *rc, double *a, double *pe)
{
    for (int i = 0; i <= P; i++) {
        double sum = 0.0;
        for (int k = 0; k < n - i; k++) sum += x[k] * x[k + i];
        r[i] = sum;
    }
    return e2o_lpca_r(P, r, rc, a, pe);
}

/* reflections -> predictor (the step-up inside lpca_r_rs.rs:26-33) -> autocorrelation
 * of the predictor polynomial ("raas"; arg name nom_raas src/ecoz2_lib/mod.rs:118) */
void e2o_ref2raas(int P, const double *rc, double *raa)
{
    double a[E2O_MAX_P + 1];
    a[0] = 1.0;
    for (int k = 1; k <= P; k++) {
        const double akk = rc[k];
        a[k] = akk;
        for (int i = 1; i <= (k >> 1); i++) {
            const double ai = a[i];
            const double aj = a[k - i];
            a[i] = ai + akk * aj;
            a[k - i] = aj + akk * ai;
        }
    }
    for (int n = 0; n <= P; n++) {
        double s = 0.0;
        for (int i = 0; i <= P - n; i++) s += a[i] * a[i + n];
        raa[n] = s;
    }
}

void e2o_codeword_q(int P, const double *raa, double *cq)
{
    cq[0] = raa[0];
    for (int n = 1; n <= P; n++) cq[n] = 2.0 * raa[n];
}

void e2o_reflections_to_cq(int P, int M, const double *reflections, double *cq)
{
    double raa[E2O_MAX_P + 1];
    for (int m = 0; m < M; m++) {
        e2o_ref2raas(P, reflections + (size_t)m * (P + 1), raa);
        e2o_codeword_q(P, raa, cq + (size_t)m * (P + 1));
    }
}

double e2o_distortion(int P, const double *r, const double *cq)
{
    double acc = E2O_FMA(r[0], cq[0], 0.0);
    for (int n = 1; n <= P; n++) acc = E2O_FMA(r[n], cq[n], acc);
    return acc;
}

/* ------------------------------------------------------------------------ */
/* fixed point                                                              */
/* ------------------------------------------------------------------------ */

void e2o_fix(double x, int sh, int64_t *hi, int64_t *lo)
{
    const double y = ldexp(x, sh);
    const double h = rint(y);
    const double l = rint(ldexp(y - h, 31));
    *hi = (int64_t)h;
    *lo = (int64_t)l;
}

/* signed 128-bit integer -> nearest double, ties to even */
static double i128_to_double(i128 v)
{
    if (v == 0) return 0.0;
    const int neg = v < 0;
    u128 u = neg ? (u128)0 - (u128)v : (u128)v;
    int msb = 127;
    while (!((u >> msb) & 1)) msb--;
    double res;
    if (msb <= 52) {
        res = (double)(uint64_t)u;
    } else {
        const int shift = msb - 52;
        uint64_t mant = (uint64_t)(u >> shift);
        const u128 rem = u & (((u128)1 << shift) - 1);
        const u128 half = (u128)1 << (shift - 1);
        if (rem > half || (rem == half && (mant & 1))) mant++;
        res = ldexp((double)mant, shift); /* mant <= 2^53: exact */
    }
    return neg ? -res : res;
}

double e2o_unfix(int64_t sum_hi, int64_t sum_lo, int sh)
{
    const i128 total = (i128)sum_hi * ((i128)1 << 31) + (i128)sum_lo;
    return ldexp(i128_to_double(total), -(sh + 31));
}

int e2o_shift_frames(double maxabs) { return 29 - ilogb(maxabs); }
int e2o_shift_frames_sq(double maxabs) { return 28 - 2 * ilogb(maxabs); }

int e2o_dist_exponent(int P, const double *cq, int M, double maxabs)
{
    double L = 0.0;
    for (int m = 0; m < M; m++) {
        const double *c = cq + (size_t)m * (P + 1);
        double s = 0.0;
        for (int n = 0; n <= P; n++) s += fabs(c[n]);
        if (s > L) L = s;
    }
    const double B = maxabs * L + 1.0;
    return ilogb(B) + 2;
}

int e2o_row_stride(int P) { return (2 * (P + 1) + 5 + 7) & ~7; }

/* ------------------------------------------------------------------------ */
/* passes                                                                   */
/* ------------------------------------------------------------------------ */

/* argmin over codewords, ascending index, strict '<' (lowest index wins ties).
 * Codebook is used transposed [n][m] in groups of E2O_GROUP so the compiler can keep
 * that many independent (vectorised) chains in flight; each (frame, codeword) chain is
 * still the canonical sequential one. */
#define E2O_GROUP 32
#define E2O_FRAME_BLOCK 64
/* Frames are taken in blocks so a codeword group (37 x 32 doubles, L1 sized) is reused across
 * the whole block; per (frame, codeword) the arithmetic and the codeword order are unchanged. */
static void assign_frames(int P, const double *cqT, int M, int Mp, const double *frames,
                          int64_t t0, int64_t t1, uint16_t *sym, double *dmin)
{
    for (int64_t tb = t0; tb < t1; tb += E2O_FRAME_BLOCK) {
        const int nb = (int)((t1 - tb) < E2O_FRAME_BLOCK ? (t1 - tb) : E2O_FRAME_BLOCK);
        double best[E2O_FRAME_BLOCK];
        int bi[E2O_FRAME_BLOCK];
        for (int i = 0; i < nb; i++) {
            best[i] = INFINITY;
            bi[i] = 0;
        }
        for (int m0 = 0; m0 < M; m0 += E2O_GROUP) {
            const int lim = (M - m0) < E2O_GROUP ? (M - m0) : E2O_GROUP;
            for (int i = 0; i < nb; i++) {
                const double *r = frames + (size_t)(tb + i) * (P + 1);
                double d[E2O_GROUP];
                const double r0 = r[0];
                for (int j = 0; j < E2O_GROUP; j++) d[j] = E2O_FMA(r0, cqT[m0 + j], 0.0);
                for (int n = 1; n <= P; n++) {
                    const double rn = r[n];
                    const double *c = cqT + (size_t)n * Mp + m0;
                    for (int j = 0; j < E2O_GROUP; j++) d[j] = E2O_FMA(rn, c[j], d[j]);
                }
                double b = best[i];
                int k = bi[i];
                for (int j = 0; j < lim; j++) {
                    if (d[j] < b) {
                        b = d[j];
                        k = m0 + j;
                    }
                }
                best[i] = b;
                bi[i] = k;
            }
        }
        for (int i = 0; i < nb; i++) {
            if (sym) sym[tb + i] = (uint16_t)bi[i];
            if (dmin) dmin[tb + i] = best[i];
        }
    }
}

static double *transpose_cq(int P, const double *cq, int M, int *Mp_out)
{
    const int Mp = (M + E2O_GROUP - 1) & ~(E2O_GROUP - 1);
    double *cqT = (double *)calloc((size_t)(P + 1) * Mp, sizeof(double));
    for (int m = 0; m < M; m++)
        for (int n = 0; n <= P; n++) cqT[(size_t)n * Mp + m] = cq[(size_t)m * (P + 1) + n];
    *Mp_out = Mp;
    return cqT;
}

void e2o_quantize(int P, const double *cq, int M, const double *frames, int64_t T, uint16_t *sym,
                  double *dmin)
{
    int Mp;
    double *cqT = transpose_cq(P, cq, M, &Mp);
    const int64_t CH = E2O_FRAME_BLOCK;
    const int64_t nch = (T + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < nch; c++) {
        const int64_t a = c * CH, b = (a + CH < T) ? a + CH : T;
        assign_frames(P, cqT, M, Mp, frames, a, b, sym, dmin);
    }
    free(cqT);
}

void e2o_pass(int P, const double *cq, int M, const double *frames, int64_t T, int sh_r, int Ed,
              uint16_t *sym, double *dmin, int64_t *rows)
{
    const int stride = e2o_row_stride(P);
    const int NC = P + 1;
    const int sh_d = 30 - Ed, sh_d2 = 30 - 2 * Ed;
    uint16_t *s = sym ? sym : (uint16_t *)malloc((size_t)T * sizeof(uint16_t));
    double *d = dmin ? dmin : (double *)malloc((size_t)T * sizeof(double));
    e2o_quantize(P, cq, M, frames, T, s, d);
    memset(rows, 0, (size_t)M * stride * sizeof(int64_t));
    /* integer accumulation: exact, so the order is immaterial */
    for (int64_t t = 0; t < T; t++) {
        int64_t *row = rows + (size_t)s[t] * stride;
        const double *r = frames + (size_t)t * NC;
        int64_t hi, lo;
        for (int n = 0; n < NC; n++) {
            e2o_fix(r[n], sh_r, &hi, &lo);
            row[2 * n] += hi;
            row[2 * n + 1] += lo;
        }
        row[2 * NC] += 1;
        const double e = d[t] - 1.0;
        e2o_fix(e, sh_d, &hi, &lo);
        row[2 * NC + 1] += hi;
        row[2 * NC + 2] += lo;
        e2o_fix(e * e, sh_d2, &hi, &lo);
        row[2 * NC + 3] += hi;
        row[2 * NC + 4] += lo;
    }
    if (!sym) free(s);
    if (!dmin) free(d);
}

int e2o_data_stats(int P, const double *frames, int64_t T, e2o_stats *st)
{
    const int NC = P + 1;
    double maxabs = 0.0;
    for (int64_t i = 0; i < T * NC; i++) {
        const double v = fabs(frames[i]);
        if (!(v <= DBL_MAX)) return 1; /* NaN or inf */
        if (v > maxabs) maxabs = v;
    }
    if (!(maxabs > 0.0)) return 2;
    memset(st, 0, sizeof *st);
    st->maxabs = maxabs;
    const int sh_r = e2o_shift_frames(maxabs), sh_q = e2o_shift_frames_sq(maxabs);
    for (int64_t t = 0; t < T; t++) {
        const double *r = frames + (size_t)t * NC;
        int64_t hi, lo;
        for (int n = 0; n < NC; n++) {
            e2o_fix(r[n], sh_r, &hi, &lo);
            st->sum_hi[n] += hi;
            st->sum_lo[n] += lo;
            e2o_fix(r[n] * r[n], sh_q, &hi, &lo);
            st->q_hi += hi;
            st->q_lo += lo;
        }
    }
    return 0;
}

void e2o_rows_stats(int P, int M, const int64_t *rows, int64_t T, int sh_r, int Ed, double Q,
                    e2o_level_stats *out)
{
    const int stride = e2o_row_stride(P);
    const int NC = P + 1;
    const int sh_d = 30 - Ed, sh_d2 = 30 - 2 * Ed;
    int64_t dh = 0, dl = 0, qh = 0, ql = 0, empty = 0;
    double within = 0.0;
    for (int m = 0; m < M; m++) {
        const int64_t *row = rows + (size_t)m * stride;
        dh += row[2 * NC + 1];
        dl += row[2 * NC + 2];
        qh += row[2 * NC + 3];
        ql += row[2 * NC + 4];
        const int64_t cnt = row[2 * NC];
        if (cnt == 0) {
            empty++;
            continue;
        }
        double ss = 0.0;
        for (int n = 0; n < NC; n++) {
            const double S = e2o_unfix(row[2 * n], row[2 * n + 1], sh_r);
            ss += S * S;
        }
        within += ss / (double)cnt;
    }
    const double DD = e2o_unfix(dh, dl, sh_d);
    const double SS = e2o_unfix(qh, ql, sh_d2);
    const double avg = DD / (double)T;
    const double q = SS / (double)T;
    const double p = avg * avg;
    double v = q - p;
    if (!(v > 0.0)) v = 0.0;
    out->DD = DD;
    out->avg = avg;
    out->sigma = sqrt(v);
    out->inertia = Q - within;
    out->empty_cells = empty;
    out->failed_cells = 0;
}

void e2o_update(int P, int M, const int64_t *rows, int sh_r, double *reflections,
                e2o_level_stats *out)
{
    const int stride = e2o_row_stride(P);
    const int NC = P + 1;
    int64_t failed = 0;
    double S[E2O_MAX_P + 1], rc[E2O_MAX_P + 1], a[E2O_MAX_P + 1], pe;
    for (int m = 0; m < M; m++) {
        const int64_t *row = rows + (size_t)m * stride;
        if (row[2 * NC] == 0) continue; /* empty cell: codeword kept (notes.md:149) */
        for (int n = 0; n < NC; n++) S[n] = e2o_unfix(row[2 * n], row[2 * n + 1], sh_r);
        if (e2o_lpca_r(P, S, rc, a, &pe) != 0) {
            failed++;
            continue; /* degenerate cell: codeword kept */
        }
        double *dst = reflections + (size_t)m * NC;
        dst[0] = 0.0;
        for (int n = 1; n <= P; n++) dst[n] = rc[n];
    }
    if (out) out->failed_cells = failed;
}

void e2o_grow(int P, int M, const double *reflections, double *grown)
{
    const int NC = P + 1;
    for (int i = 0; i < M; i++) {
        const double *src = reflections + (size_t)i * NC;
        double *d0 = grown + (size_t)(2 * i) * NC;
        double *d1 = grown + (size_t)(2 * i + 1) * NC;
        d0[0] = 0.0;
        d1[0] = 0.0;
        for (int n = 1; n <= P; n++) {
            d0[n] = src[n] * 0.99;
            d1[n] = src[n] * 1.01;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* files                                                                    */
/* ------------------------------------------------------------------------ */

static int mkdirs_for(const char *path)
{
    char buf[4096];
    snprintf(buf, sizeof buf, "%s", path);
    for (char *p = buf + 1; *p; p++) {
        if (*p == '/') {
            *p = 0;
            if (mkdir(buf, 0777) != 0 && errno != EEXIST) return -1;
            *p = '/';
        }
    }
    return 0;
}

static void put_u32(FILE *f, uint32_t v)
{
    unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16),
                          (unsigned char)(v >> 24)};
    fwrite(b, 1, 4, f);
}

static int get_u32(FILE *f, uint32_t *v)
{
    unsigned char b[4];
    if (fread(b, 1, 4, f) != 4) return -1;
    *v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    return 0;
}

static void put_header(FILE *f, const char *ident, const char *class_name)
{
    char id[E2O_FILE_IDENT_LEN] = {0}, cn[E2O_MAX_CLASS_NAME_LEN] = {0};
    strncpy(id, ident, E2O_FILE_IDENT_LEN - 1);
    strncpy(cn, class_name, E2O_MAX_CLASS_NAME_LEN - 1);
    fwrite(id, 1, sizeof id, f);
    fwrite(cn, 1, sizeof cn, f);
}

static int get_header(FILE *f, const char *ident, char class_name[E2O_MAX_CLASS_NAME_LEN])
{
    char id[E2O_FILE_IDENT_LEN];
    if (fread(id, 1, sizeof id, f) != sizeof id) return -1;
    if (strncmp(id, ident, strlen(ident)) != 0) return -2;
    if (fread(class_name, 1, E2O_MAX_CLASS_NAME_LEN, f) != E2O_MAX_CLASS_NAME_LEN) return -1;
    class_name[E2O_MAX_CLASS_NAME_LEN - 1] = 0;
    return 0;
}

/* .prd: 16B ident "<predictor>", 96B class, u32 T, u32 P, T*(P+1) f64 LE */
int e2o_prd_save(const char *path, const char *class_name, int P, const double *frames, int64_t T)
{
    if (mkdirs_for(path) != 0) return -1;
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    put_header(f, "<predictor>", class_name);
    put_u32(f, (uint32_t)T);
    put_u32(f, (uint32_t)P);
    fwrite(frames, sizeof(double), (size_t)T * (P + 1), f);
    return fclose(f);
}

int e2o_prd_load(const char *path, char class_name[E2O_MAX_CLASS_NAME_LEN], int *P, double **frames,
                 int64_t *T)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint32_t t, p;
    if (get_header(f, "<predictor>", class_name) || get_u32(f, &t) || get_u32(f, &p)) {
        fclose(f);
        return -2;
    }
    double *buf = (double *)malloc((size_t)t * (p + 1) * sizeof(double) + 8);
    if (fread(buf, sizeof(double), (size_t)t * (p + 1), f) != (size_t)t * (p + 1)) {
        free(buf);
        fclose(f);
        return -3;
    }
    fclose(f);
    *P = (int)p;
    *T = (int64_t)t;
    *frames = buf;
    return 0;
}

/* .cbook: 16B ident "<codebook>", 96B class, u32 P, u32 M, M*(P+1) f64 reflections (F4) */
int e2o_cbook_save(const char *path, const char *class_name, int P, int M, const double *reflections)
{
    if (mkdirs_for(path) != 0) return -1;
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    put_header(f, "<codebook>", class_name);
    put_u32(f, (uint32_t)P);
    put_u32(f, (uint32_t)M);
    fwrite(reflections, sizeof(double), (size_t)M * (P + 1), f);
    return fclose(f);
}

int e2o_cbook_load(const char *path, char class_name[E2O_MAX_CLASS_NAME_LEN], int *P, int *M,
                   double **reflections)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint32_t p, m;
    if (get_header(f, "<codebook>", class_name) || get_u32(f, &p) || get_u32(f, &m)) {
        fclose(f);
        return -2;
    }
    double *buf = (double *)malloc((size_t)m * (p + 1) * sizeof(double) + 8);
    if (fread(buf, sizeof(double), (size_t)m * (p + 1), f) != (size_t)m * (p + 1)) {
        free(buf);
        fclose(f);
        return -3;
    }
    fclose(f);
    *P = (int)p;
    *M = (int)m;
    *reflections = buf;
    return 0;
}

/* .seq: src/sequence/mod.rs:49-75 -- ident "<sequence>", class, u32 T, u32 M, T*u16 */
int e2o_seq_save(const char *path, const char *class_name, int M, const uint16_t *sym, int64_t T)
{
    if (mkdirs_for(path) != 0) return -1;
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    put_header(f, "<sequence>", class_name);
    put_u32(f, (uint32_t)T);
    put_u32(f, (uint32_t)M);
    for (int64_t t = 0; t < T; t++) {
        unsigned char b[2] = {(unsigned char)sym[t], (unsigned char)(sym[t] >> 8)};
        fwrite(b, 1, 2, f);
    }
    return fclose(f);
}

int e2o_seq_load(const char *path, char class_name[E2O_MAX_CLASS_NAME_LEN], int *M, uint16_t **sym,
                 int64_t *T)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint32_t t, m;
    if (get_header(f, "<sequence>", class_name) || get_u32(f, &t) || get_u32(f, &m)) {
        fclose(f);
        return -2;
    }
    uint16_t *buf = (uint16_t *)malloc((size_t)t * 2 + 8);
    for (uint32_t i = 0; i < t; i++) {
        unsigned char b[2];
        if (fread(b, 1, 2, f) != 2) {
            free(buf);
            fclose(f);
            return -3;
        }
        buf[i] = (uint16_t)(b[0] | (b[1] << 8));
    }
    fclose(f);
    *M = (int)m;
    *T = (int64_t)t;
    *sym = buf;
    return 0;
}

void e2o_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* LBG driver (shape: notes.md:122-153)                         */
/* ------------------------------------------------------------------------ */

int e2o_learn(int P, double eps, const char *class_name, const double *frames, int64_t T,
              const double *base_reflections, int base_M, int max_M, const char *out_root,
              void *target, e2o_learn_cb cb, void *hook_user, e2o_level_hook hook)
{
    if (P < 1 || P > E2O_MAX_P || T < 1) return 1;
    const int NC = P + 1;
    const int stride = e2o_row_stride(P);
    e2o_stats st;
    if (e2o_data_stats(P, frames, T, &st) != 0) return 2;
    const int sh_r = e2o_shift_frames(st.maxabs), sh_q = e2o_shift_frames_sq(st.maxabs);
    const double Q = e2o_unfix(st.q_hi, st.q_lo, sh_q);

    int M;
    double *refl = (double *)calloc((size_t)max_M * 2 * NC, sizeof(double));
    double *grown = (double *)calloc((size_t)max_M * 2 * NC, sizeof(double));
    double *cq = (double *)calloc((size_t)max_M * 2 * NC, sizeof(double));
    int64_t *rows = (int64_t *)calloc((size_t)max_M * 2 * stride, sizeof(int64_t));
    int rc_status = 0;

    if (base_reflections) {
        M = base_M;
        memcpy(refl, base_reflections, (size_t)M * NC * sizeof(double));
    } else {
        /* M = 1: centroid of the whole training set */
        double S[E2O_MAX_P + 1], rc[E2O_MAX_P + 1], a[E2O_MAX_P + 1], pe;
        for (int n = 0; n < NC; n++) S[n] = e2o_unfix(st.sum_hi[n], st.sum_lo[n], sh_r);
        if (e2o_lpca_r(P, S, rc, a, &pe) != 0) {
            rc_status = 3;
            goto done;
        }
        M = 1;
        refl[0] = 0.0;
        for (int n = 1; n <= P; n++) refl[n] = rc[n];
    }

    FILE *rpt = NULL;
    char path[4096];
    if (out_root) {
        snprintf(path, sizeof path, "%s/data/codebooks/%s/eps_%g.rpt", out_root, class_name, eps);
        if (mkdirs_for(path) == 0) rpt = fopen(path, "w");
        if (rpt) fprintf(rpt, "# %lld training vectors, P=%d, eps=%g\n# M passes DD avg_distortion sigma inertia empty_cells\n",
                         (long long)T, P, eps);
    }

    double DDprv = DBL_MAX / 1e5; /* "e+303" in notes.md:128 */
    while (M < max_M) {
        e2o_grow(P, M, refl, grown);
        M *= 2;
        memcpy(refl, grown, (size_t)M * NC * sizeof(double));

        e2o_level_stats ls;
        int pass = 0;
        for (;; pass++) {
            e2o_reflections_to_cq(P, M, refl, cq);
            const int Ed = e2o_dist_exponent(P, cq, M, st.maxabs);
            e2o_pass(P, cq, M, frames, T, sh_r, Ed, NULL, NULL, rows);
            e2o_rows_stats(P, M, rows, T, sh_r, Ed, Q, &ls);
            const double DD = ls.DD;
            /* notes.md:128-153: pass index starts at 0 and pass 0 never terminates a level;
             * DDprv carries over from the previous level */
            /* safety cap of 1000 passes per level (eps <= 0 would never terminate) */
            const int converged = (pass > 0 && !(((DDprv - DD) / DD) >= eps)) || pass + 1 >= 1000;
            DDprv = DD;
            if (converged) break;
            e2o_update(P, M, rows, sh_r, refl, &ls);
        }
        if (out_root) {
            snprintf(path, sizeof path, "%s/data/codebooks/%s/eps_%g_M_%04d.cbook", out_root,
                     class_name, eps, M);
            e2o_cbook_save(path, class_name, P, M, refl);
            if (rpt)
                fprintf(rpt, "%d %d %.17g %.17g %.17g %.17g %lld\n", M, pass + 1, ls.DD, ls.avg,
                        ls.sigma, ls.inertia, (long long)ls.empty_cells);
        }
        if (hook) hook(hook_user, M, pass + 1, refl, &ls);
        if (cb) cb(target, M, ls.avg, ls.sigma, ls.inertia);
    }
    if (rpt) fclose(rpt);
done:
    free(refl);
    free(grown);
    free(cq);
    free(rows);
    return rc_status;
}

/* ------------------------------------------------------------------------ */

void e2o_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

double e2o_time_pass(int P, const double *cq, int M, const double *frames, int64_t T, int reps,
                     int *threads_used)
{
    uint16_t *sym = (uint16_t *)malloc((size_t)T * 2);
    double *dmin = (double *)malloc((size_t)T * 8);
#ifdef _OPENMP
    *threads_used = omp_get_max_threads();
#else
    *threads_used = 1;
#endif
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int i = 0; i < reps; i++) e2o_quantize(P, cq, M, frames, T, sym, dmin);
    clock_gettime(CLOCK_MONOTONIC, &b);
    free(sym);
    free(dmin);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
