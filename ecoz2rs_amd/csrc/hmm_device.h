// hmm_device.h -- device-side interface of the HMM kernels (internal; the C-ABI is include/ecoz2_classify.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace e2hmm {

constexpr int WAVE_N = 64;     // up to here one wavefront per sequence, one lane per state (k_hmm_score, k_hmm_fb)
constexpr int MAX_N = 512;     // beyond: one workgroup per sequence, one thread per state (k_hmm_score_wg, k_hmm_fb_wg)
constexpr int ACC_SHIFT = 29;  // expected counts are < 2: x ~= (hi*2^31 + lo) * 2^-(29+31)  (oracle: E2H_ACC_SHIFT)

struct ModelDev {
    int N, M;
    const double* pi;  // [N]
    const double* A;   // [N][N], row = from-state
    const double* B;   // [N][M], row = state
};

// int64 words of the E-step accumulators: [hi, lo] pairs  PI[N] | AN[N][N] | AD[N] | BN[N][M] | BD[N] | used, skipped
long long acc_words(int N, int M);

// scaled forward pass of every (sequence, model) pair; results at [s * K + k]: P(O) = mant * 2^exp2, status 0 ok,
// 1 the model cannot emit the sequence, 2 symbol outside the model's alphabet.  offs: S+1 symbol offsets.
void launch_score(const ModelDev* models, int K, int maxN, const unsigned short* sym, const long long* offs, int S,
                  double* mant, long long* exp2, int* status, hipStream_t st);
// Baum-Welch E-step of one model over S sequences (expected counts added to acc; per-sequence P(O) and status out).
// alpha_buf: total_symbols x N doubles, c_buf: total_symbols doubles (scratch).
// scratch (N > WAVE_N only; may be null otherwise): fb_scratch_words(N) int64 words, contents irrelevant
void launch_fb(const ModelDev& md, const unsigned short* sym, const long long* offs, int S, double* alpha_buf,
               double* c_buf, long long* acc, double* mant, long long* exp2, int* status, hipStream_t st,
               long long* scratch = nullptr);
long long fb_scratch_words(int N);
// M-step (+ epsilon restriction on B when epsilon > 0), in place
void launch_reestimate(int N, int M, const long long* acc, double epsilon, double* pi, double* A, double* B,
                       hipStream_t st);

}  // namespace e2hmm
