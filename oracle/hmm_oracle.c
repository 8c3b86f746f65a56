/* hmm_oracle.c -- see hmm_oracle.h.  TEST INFRASTRUCTURE ONLY; strict IEEE (-ffp-contract=off, explicit fma()). */
#include "hmm_oracle.h"

#include "vq_oracle.h" /* e2o_fix / e2o_unfix: the exact fixed-point scheme of the VQ accumulate */

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- generator ---------------------------------------------------------------------------- */
static uint64_t g_state = 0x9E3779B97F4A7C15ull;

static uint64_t splitmix64_next(void)
{
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint64_t e2h_set_random_seed(int64_t seed)
{
    const uint64_t s = seed < 0 ? (uint64_t)time(NULL) : (uint64_t)seed;
    g_state = s;
    return s;
}

double e2h_uniform(void) { return (double)(splitmix64_next() >> 11) * 0x1.0p-53; }

/* ---- model ------------------------------------------------------------------------------------ */
static void random_row(double *row, int n)
{
    double s = 0.0;
    for (int k = 0; k < n; k++) {
        /* strictly positive draws: (x + 1) * 2^-53 in (0, 1] */
        row[k] = (double)((splitmix64_next() >> 11) + 1) * 0x1.0p-53;
        s = s + row[k];
    }
    for (int k = 0; k < n; k++) row[k] = row[k] / s;
}

int e2h_init(int N, int M, int type, double *pi, double *A, double *B)
{
    if (N < 1 || N > E2H_MAX_N || M < 1 || M > 65536 || type < 0 || type > 3) return 1;
    if (type == 0) {
        random_row(pi, N);
        for (int i = 0; i < N; i++) random_row(A + (size_t)i * N, N);
        for (int j = 0; j < N; j++) random_row(B + (size_t)j * M, M);
    } else if (type == 1) {
        for (int i = 0; i < N; i++) pi[i] = 1.0 / (double)N;
        for (int i = 0; i < N * N; i++) A[i] = 1.0 / (double)N;
        for (size_t k = 0; k < (size_t)N * M; k++) B[k] = 1.0 / (double)M;
    } else {
        const int width = type == 2 ? 2 : 3; /* cascade-2: i -> i, i+1; cascade-3: i -> i, i+1, i+2 */
        for (int i = 0; i < N; i++) pi[i] = i == 0 ? 1.0 : 0.0;
        for (int i = 0; i < N; i++) {
            const int reach = (N - i) < width ? (N - i) : width;
            for (int j = 0; j < N; j++) A[(size_t)i * N + j] = (j >= i && j < i + reach) ? 1.0 / (double)reach : 0.0;
        }
        for (int j = 0; j < N; j++) random_row(B + (size_t)j * M, M);
    }
    return 0;
}

/* ---- scoring ---------------------------------------------------------------------------------- */
int e2h_forward(int N, int M, const double *pi, const double *A, const double *B, const uint16_t *o, int64_t T,
                double *mant, int64_t *exp2, double *alpha_hat, double *c_out)
{
    double al[E2H_MAX_N], nx[E2H_MAX_N];
    double p = 0.5; /* P = p * 2^E, p in [0.5, 1) */
    int64_t E = 1;
    for (int64_t t = 0; t < T; t++) {
        const int ot = o[t];
        if (ot >= M) return 2;
        if (t == 0) {
            for (int j = 0; j < N; j++) nx[j] = pi[j] * B[(size_t)j * M + ot];
        } else {
            for (int j = 0; j < N; j++) {
                double acc = 0.0;
                for (int i = 0; i < N; i++) acc = fma(al[i], A[(size_t)i * N + j], acc);
                nx[j] = acc * B[(size_t)j * M + ot];
            }
        }
        double c = 0.0;
        for (int j = 0; j < N; j++) c = c + nx[j];
        if (!(c > 0.0)) {
            *mant = 0.0;
            *exp2 = 0;
            return 1;
        }
        for (int j = 0; j < N; j++) al[j] = nx[j] / c;
        if (alpha_hat) memcpy(alpha_hat + (size_t)t * N, al, (size_t)N * sizeof(double));
        if (c_out) c_out[t] = c;
        int e, e2;
        const double m = frexp(c, &e);
        p = frexp(p * m, &e2);
        E += (int64_t)e + (int64_t)e2;
    }
    *mant = p;
    *exp2 = E;
    return 0;
}

double e2h_log_prob(double mant, int64_t exp2)
{
    if (!(mant > 0.0)) return -INFINITY;
    return log(mant) + (double)exp2 * M_LN2;
}

/* ---- training ------------------------------------------------------------------------------- */
int64_t e2h_acc_words(int N, int M) { return 2 * ((int64_t)N + (int64_t)N * N + N + (int64_t)N * M + N) + 2; }

static void acc_add(int64_t *cell, double x)
{
    int64_t hi, lo;
    e2o_fix(x, E2H_ACC_SHIFT, &hi, &lo);
    cell[0] += hi;
    cell[1] += lo;
}

int e2h_accumulate(int N, int M, const double *pi, const double *A, const double *B, const uint16_t *o, int64_t T,
                   int64_t *acc, double *mant, int64_t *exp2)
{
    int64_t *PI = acc, *AN = PI + 2 * N, *AD = AN + 2 * (int64_t)N * N, *BN = AD + 2 * N, *BD = BN + 2 * (int64_t)N * M;
    int64_t *counts = BD + 2 * N;
    if (T < 1) {
        counts[1] += 1;
        *mant = 0.5;
        *exp2 = 1;
        return 1;
    }
    double *alpha = (double *)malloc((size_t)T * N * sizeof(double));
    double *c = (double *)malloc((size_t)T * sizeof(double));
    const int st = e2h_forward(N, M, pi, A, B, o, T, mant, exp2, alpha, c);
    if (st != 0) {
        counts[1] += 1;
        free(alpha);
        free(c);
        return st;
    }
    double beta[E2H_MAX_N], u[E2H_MAX_N], nb[E2H_MAX_N];
    for (int i = 0; i < N; i++) beta[i] = 1.0;
    for (int64_t t = T - 1; t >= 0; t--) {
        const double *al = alpha + (size_t)t * N;
        if (t < T - 1) {
            /* beta holds beta^_{t+1}: u, then xi_t and beta^_t */
            const int o1 = o[t + 1];
            for (int j = 0; j < N; j++) u[j] = (B[(size_t)j * M + o1] * beta[j]) / c[t + 1];
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++) acc_add(AN + 2 * ((int64_t)i * N + j), (al[i] * A[(size_t)i * N + j]) * u[j]);
            for (int i = 0; i < N; i++) {
                double a = 0.0;
                for (int j = 0; j < N; j++) a = fma(A[(size_t)i * N + j], u[j], a);
                nb[i] = a;
            }
            memcpy(beta, nb, (size_t)N * sizeof(double));
        }
        for (int i = 0; i < N; i++) {
            const double g = al[i] * beta[i];
            if (t < T - 1) acc_add(AD + 2 * i, g);
            acc_add(BN + 2 * ((int64_t)i * M + o[t]), g);
            acc_add(BD + 2 * i, g);
            if (t == 0) acc_add(PI + 2 * i, g);
        }
    }
    counts[0] += 1;
    free(alpha);
    free(c);
    return 0;
}

void e2h_reestimate(int N, int M, const int64_t *acc, double epsilon, double *pi, double *A, double *B)
{
    const int64_t *PI = acc, *AN = PI + 2 * N, *AD = AN + 2 * (int64_t)N * N, *BN = AD + 2 * N, *BD = BN + 2 * (int64_t)N * M;
    const int64_t used = BD[2 * N];
    if (used > 0)
        for (int i = 0; i < N; i++) pi[i] = e2o_unfix(PI[2 * i], PI[2 * i + 1], E2H_ACC_SHIFT) / (double)used;
    for (int i = 0; i < N; i++) {
        const double den = e2o_unfix(AD[2 * i], AD[2 * i + 1], E2H_ACC_SHIFT);
        if (den > 0.0)
            for (int j = 0; j < N; j++) {
                const int64_t *cell = AN + 2 * ((int64_t)i * N + j);
                A[(size_t)i * N + j] = e2o_unfix(cell[0], cell[1], E2H_ACC_SHIFT) / den;
            }
    }
    for (int j = 0; j < N; j++) {
        double *row = B + (size_t)j * M;
        const double den = e2o_unfix(BD[2 * j], BD[2 * j + 1], E2H_ACC_SHIFT);
        if (den > 0.0)
            for (int k = 0; k < M; k++) {
                const int64_t *cell = BN + 2 * ((int64_t)j * M + k);
                row[k] = e2o_unfix(cell[0], cell[1], E2H_ACC_SHIFT) / den;
            }
        if (epsilon > 0.0) { /* hmm_adjustb: floor, then renormalise */
            double s = 0.0;
            for (int k = 0; k < M; k++) {
                if (row[k] < epsilon) row[k] = epsilon;
                s = s + row[k];
            }
            for (int k = 0; k < M; k++) row[k] = row[k] / s;
        }
    }
}

int e2h_learn(int N, int M, const uint16_t *const *seqs, const int64_t *lens, int R, double epsilon, double val_auto,
              int max_iterations, double *pi, double *A, double *B, double *sum_log_prob, int cap,
              e2h_learn_callback_t callback)
{
    const int64_t W = e2h_acc_words(N, M);
    int64_t *acc = (int64_t *)malloc((size_t)W * sizeof(int64_t));
    if (!acc) return -1;
    int it = 0;
    double Lprev = 0.0;
    static char var[] = "sum_log_prob";
    for (;;) {
        if ((max_iterations >= 0 && it >= max_iterations) || it >= E2H_MAX_ESTEPS) break;
        memset(acc, 0, (size_t)W * sizeof(int64_t));
        double L = 0.0;
        for (int r = 0; r < R; r++) {
            double mant;
            int64_t e2;
            if (e2h_accumulate(N, M, pi, A, B, seqs[r], lens[r], acc, &mant, &e2) == 0) L = L + e2h_log_prob(mant, e2);
        }
        if (it < cap) sum_log_prob[it] = L;
        if (callback) callback(var, L);
        if (it > 0 && L - Lprev <= val_auto) {
            it++;
            break;
        }
        e2h_reestimate(N, M, acc, epsilon, pi, A, B);
        Lprev = L;
        it++;
    }
    free(acc);
    return it;
}

/* ---- files ---------------------------------------------------------------------------------- */
int e2h_save(const char *path, const char *class_name, int N, int M, const double *pi, const double *A, const double *B)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    char ident[E2O_FILE_IDENT_LEN] = "<hmm>", cls[E2O_MAX_CLASS_NAME_LEN];
    memset(cls, 0, sizeof cls);
    strncpy(cls, class_name, sizeof cls - 1);
    const uint32_t hdr[2] = {(uint32_t)N, (uint32_t)M}; /* little-endian host assumed, as in vq_oracle.c */
    int ok = fwrite(ident, 1, sizeof ident, f) == sizeof ident && fwrite(cls, 1, sizeof cls, f) == sizeof cls &&
             fwrite(hdr, 4, 2, f) == 2 && fwrite(pi, 8, (size_t)N, f) == (size_t)N &&
             fwrite(A, 8, (size_t)N * N, f) == (size_t)N * N && fwrite(B, 8, (size_t)N * M, f) == (size_t)N * M;
    return (fclose(f) != 0 || !ok) ? 1 : 0;
}

int e2h_load_info(const char *path, char class_name[96], int *N, int *M)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    char ident[E2O_FILE_IDENT_LEN];
    uint32_t hdr[2];
    const int ok = fread(ident, 1, sizeof ident, f) == sizeof ident && strncmp(ident, "<hmm>", 5) == 0 &&
                   fread(class_name, 1, 96, f) == 96 && fread(hdr, 4, 2, f) == 2;
    fclose(f);
    if (!ok) return 1;
    class_name[95] = 0;
    *N = (int)hdr[0];
    *M = (int)hdr[1];
    return 0;
}

int e2h_load(const char *path, double *pi, double *A, double *B)
{
    char cls[96];
    int N, M;
    if (e2h_load_info(path, cls, &N, &M)) return 1;
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    fseek(f, 16 + 96 + 8, SEEK_SET);
    const int ok = fread(pi, 8, (size_t)N, f) == (size_t)N && fread(A, 8, (size_t)N * N, f) == (size_t)N * N &&
                   fread(B, 8, (size_t)N * M, f) == (size_t)N * M;
    fclose(f);
    return ok ? 0 : 1;
}
