// vq_device.h -- device-side interface of the VQ kernels (internal; the C-ABI is include/ecoz2_vq.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define E2VQ_MAX_P 200          // CHANGELOG.md:183 of the reference: "increased maximum prediction order (200)"
#define E2VQ_LDS_BYTES 163840   // 160 KiB per CU on gfx950
#define E2VQ_PRE_EBIAS (1 << 20)  // bias of the codebook-scale exponent kept in the prefilter's scalars (0 = empty)

namespace e2vq {

struct DevScalars {
    double maxabs;  // max |x| over the whole training set (all ranks)
    double Q;       // exact sum of squares of the training set (all ranks)
    int sh_r;       // frame fixed-point shift   29 - ilogb(maxabs)
    int sh_q;       // square fixed-point shift  28 - 2*ilogb(maxabs)
};

inline int row_stride(int NC) { return (2 * NC + 5 + 7) & ~7; }
inline int cb_pad(int NC) { return (NC + 7) & ~7; }

bool uses_mfma(int NC);            // P = 4 .. 80: the sweep runs on the FP64 matrix pipe
bool mfma_is_wide(int NC);         // P = 41 .. 80: half blocks per wave; no LDS-table accumulate, no row-major quantize sweep
int mfma_hybrid_cells(int NC);     // cells of the hybrid accumulate's LDS table
// generic orders (P > 40): scratch for the transposed codebook of k_pass_generic_lds, passed to launch_pass in the cbm slot
inline long generic_scratch_doubles(int NC, int M) { return (long)NC * ((M + 7) / 8 * 8); }
inline long cbm_doubles(int NC, int M) { return (long)((M + 15) / 16) * (((((NC + 3) / 4) + 1) / 2) * 128 + 16); }

bool launch_blockify(const double* aos, long T, int NC, int FB, double* blk, long nblocks,
                     unsigned long long* maxabs_bits, int* bad, hipStream_t s);
void launch_maxabs(const double* blk, long count, unsigned long long* out_bits, int* bad, hipStream_t s);
void launch_finish_scalars(const unsigned long long* maxabs_bits, DevScalars* sc, hipStream_t s);
void launch_global_sums(const double* blk, long nblocks, int NC, int FB, const DevScalars* sc, long long* stats,
                        hipStream_t s);
// mode 0 assign only, 1 LDS accumulators, 2 global atomics
int launch_pass(int NC, int mode, const double* blk, long T, long nblocks, const double* cbq, const double* cbm,
                int M, const DevScalars* sc, const unsigned long long* l1max_bits, unsigned short* sym, double* dmin,
                long long* rows, hipStream_t s);
// prediction orders with a prefiltered sweep (NC = P + 1): the usual LPC orders 12, 16, ..., 40
#ifndef E2VQ_PRE_NC_LIST
#define E2VQ_PRE_NC_LIST(X) X(13) X(17) X(21) X(25) X(29) X(33) X(37) X(41)
#endif
// prefiltered pass (vq_prefilter.hip): f16 limb images + exact candidate evaluation + fallback list
bool prefilter_supports(int NC, int M);
size_t prefilter_frame_image_bytes(int NC, long nblocks64);
size_t prefilter_codebook_image_bytes(int NC, int M);
size_t prefilter_scalars_bytes();
void launch_prefilter_frames(const double* blk, long T, long nblocks64, int NC, unsigned long long* colmax_bits, int* ea,
                             void* fimg, float* fg, hipStream_t s);
// quantize, unfused (P = 40): row-major frames -> (optional) blocked FP64 layout + limb image + tolerance terms, with the
// scales of launch_prefilter_quantize_scales
void launch_prefilter_quantize_prep(const double* aos, long T, long nblocks64, int NC, const int* ea, double* blk,
                                    void* fimg, float* fg, hipStream_t s);
// scale_ready: the scalars were zeroed before and *prefilter_codebook_scale(ps) already holds the scale of this
// codebook (k_cell_update computed it): only the image kernel runs
void launch_prefilter_codebook(const double* cbq, int M, int NC, const int* ea, void* ps, void* cimg, hipStream_t s,
                               bool scale_ready = false);
const int* prefilter_fallback_count(const void* ps);
int* prefilter_codebook_scale(void* ps);
int launch_pass_prefiltered(int NC, bool accumulate, const double* blk, long T, long nblocks, const void* fimg,
                            const float* fg, const void* cimg, void* ps, const double* cbq, int M,
                            const DevScalars* sc, const unsigned long long* l1max_bits, unsigned short* sym, double* dmin,
                            long long* rows, int* fb_list, unsigned short* prev_sym, bool incremental,
                            hipStream_t s, const double* rowmajor_frames = nullptr, const int* ea_fused = nullptr,
                            const double* resident_rowmajor = nullptr, long long* family_table = nullptr,
                            const struct PassRecords* records = nullptr);
// records (accumulate, resident_rowmajor): the accumulating kernel RECORDS the frames that contribute instead of adding them
// -- (frame, cell within its bin, sign) as 8 bytes into the region of (sweeping workgroup, bin of cells), positions from
// per-workgroup LDS counters -- and launch_reduce_records folds them into the rows through LDS tables, one workgroup per
// (bin, slice of the regions): work that follows the number of contributions.  Plan with prefilter_records_plan.
struct PassRecords {
    void* recs;        // grid * nbins * cap records of 8 bytes
    int* counts;       // grid * nbins
    int grid;          // workgroups of the sweep (= regions per bin)
    int nbins;         // bins of rows + bins of the family side table
    int nbins_rows;
    int bin_cells;     // cells per bin (the LDS table of a reducing workgroup)
    unsigned magic;    // cell / bin_cells == (cell * magic) >> 22 for every cell of the plan
    int cap;           // records per region
    long long* total_out;  // (optional, host-mapped) the reduce kernel stores the pass's number of records here
};
// fills everything but the two pointers; false: no plan (order / size not served).  bytes: what recs needs.
bool prefilter_records_plan(int NC, int M, bool family, long nblocks, PassRecords* plan, size_t* recs_bytes);
int launch_reduce_records(int NC, const double* aos, const PassRecords& plan, bool few, const DevScalars* sc, long long* rows,
                          long long* family_table, hipStream_t s);
// round 5 (vq_sweep.hip): the accumulating prefiltered pass as sort (once per level) + candidate sweep + finishing kernel +
// launch_reduce_records.  sweep_supported: the order has a prefiltered sweep and its rows fit the finishing kernel's LDS.
bool sweep_supported(int NC, int M);
bool sweep_fused_supported(int NC, int M);   // ... and the fused sorted pass (launch_pass_sorted) as well
size_t sweep_frame_image_bytes(int NC, long nblocks64);
// frame-major limb image (same limbs as launch_prefilter_frames, whose scales `ea` it uses) from the row-major resident copy
void launch_sweep_frames(const double* aos, long T, long nblocks64, int NC, const int* ea, void* img, hipStream_t s);
size_t sort_scratch_bytes();                 // zero it once; holds the sort's histogram / cursor and the sweep's counters
void* sweep_counters_of(void* sort_scratch); // two unsigned 64-bit words: flagged jobs, jobs (the sweeps whose share the host fetches: launch_sweep_counters_out)
void* sweep_totals_of(void* sort_scratch);   // the same pair for every other two-stage sweep (read back by e2vq_sweep_executed)
// perm[slot] = frame, frames grouped by key (< nbins <= 8192); perm holds nblocks64 * 64 slots
int launch_sort_by_cell(const unsigned short* key, long T, long nblocks64, int nbins, void* scratch, unsigned* perm, hipStream_t s);
// cand[frame] = c1 | c2 << 13 | amb << 26 | cert << 27.  perm == nullptr: slots are frames.  home_mul: 0 = no home tile,
// else the home codeword of a block is home_mul * prev_sym[its first frame] (1: cells of the previous pass, 2: of the parents)
int launch_sweep_candidates(int NC, bool two_stage, const void* fimg, const unsigned* perm, long T, long nblocks, const void* cimg,
                            const void* ps, int M, const unsigned short* prev_sym, int home_mul, unsigned* cand, void* counters,
                            hipStream_t s);
// the FUSED pass over grouped frames (perm != nullptr, incr 1 or 2): sweep + exact evaluation + outputs + the cell sums
// reduced in the block, one kernel; `cells`: every frame's cell, read as the old one and written with the new one.
// counters: sweep_counters_of(sort scratch) or nullptr (left for the host to fetch: launch_sweep_counters_out)
int launch_pass_sorted(int NC, bool two_stage, bool one_block, const void* fimg, const unsigned* perm, long T, long nblocks, const void* cimg, void* ps,
                       const double* cbq, int M, const double* aos, const DevScalars* sc, const unsigned long long* l1max_bits,
                       unsigned short* sym, double* dmin, long long* rows, long long* fam, int* fb_list, unsigned short* cells, int incr,
                       void* counters, hipStream_t s);
// copies the two-stage sweep's counters to two host-mapped 64-bit words and zeroes them (k_finish does the same for the
// split pass)
void launch_sweep_counters_out(void* counters, void* host_counters, hipStream_t s);
// incr: 0 full, 1 incremental, 2 seeded (as k_pass_pre_lds); the contributions are recorded for launch_reduce_records
int launch_finish(int NC, const double* aos, long T, long nblocks, const unsigned* cand, void* ps, const double* cbq, int M,
                  const DevScalars* sc, const unsigned long long* l1max_bits, unsigned short* sym, double* dmin, long long* rows,
                  int* fb_list, unsigned short* prev_sym, int incr, const struct PassRecords* records, void* counters,
                  void* host_counters, hipStream_t s);
// (counters / host_counters: the device words of the two-stage sweep in front -- sweep_counters_of -- and two host-mapped
// 64-bit words the kernel copies them to before it zeroes them; both null for a one-stage sweep)
// resident_rowmajor (accumulating passes): a row-major copy of the training frames padded with zero rows to whole
// 64-frame blocks; with it (and prefilter_lds_stage(NC)) the pass runs k_pass_pre_lds; prev_sym must then be padded
// by 128 bytes
bool prefilter_lds_stage(int NC);
// ... and rows of at most 80 elements (P <= 39) for its burst of atomics / the seeded first pass without records
bool prefilter_burst_supported(int NC);
// fused quantize (ea_fused != nullptr, assignment only): no frame image at all -- the sweep builds the limb images of
// its frames from rowmajor_frames with the per-coefficient scales ea_fused (launch_prefilter_quantize_scales)
bool prefilter_fused_quantize(int NC);
void launch_prefilter_quantize_scales(const double* cbq, int M, int NC, int* ea, hipStream_t s);
// incremental: 0 = full, 1 = incremental (old cell = prev_sym), 2 = the seeded first pass of a level (old cell = 2 prev_sym)
int launch_pass_fallback(int NC, bool accumulate, const double* blk, const double* cbm, int M, const DevScalars* sc,
                         const unsigned long long* l1max_bits, unsigned short* sym, double* dmin, long long* rows,
                         const int* fb_list, const int* fb_count, unsigned short* prev_sym, int incremental,
                         hipStream_t s, bool rowmajor = false, unsigned short* cells_out = nullptr, bool long_list = false);
// (long_list: the caller expects tens of per cent of the frames on the list -- the sweep then runs in the plain pass's shape)
// (cells_out: where the listed frames' new cells are recorded -- default: in place, prev_sym)
// the seeded first pass after a split (vq_update.hip: k_seed_family): rows <- parents' sums in the even children, X <- 0;
// after the pass (and its fallback sweep) launch_family_fixup moves the in-family arrivals X[i] from row 2 i to row 2 i + 1
void launch_seed_family(const long long* parent, long long* rows, long long* X, int Mold, int NC, hipStream_t s,
                        const struct ZeroList* zero = nullptr);
void launch_family_fixup(long long* rows, const long long* X, int Mold, int NC, hipStream_t s);
void launch_zero_distortion_columns(long long* rows, int M, int NC, hipStream_t s);
void launch_rows_stats(const long long* rows, int M, int NC, const DevScalars* sc, double* S, double* within,
                       long long* lstats, hipStream_t s);
void launch_centroids(const long long* rows, const double* S, int M, int NC, const double* refl_in, double* refl_out,
                      long long* lstats, hipStream_t s);
// in-process multi-GPU exchange: the caller's slice [lo, hi) of `count` words is combined over the n ranks' buffers
// (peer pointers) and written back to all of them
constexpr int E2VQ_MAX_LOCAL_RANKS = 16;
struct PeerBuffers {
    long long* p[E2VQ_MAX_LOCAL_RANKS];
};
void launch_reduce_slice_i64(const PeerBuffers& bufs, int n, long lo, long hi, int op, hipStream_t s);
// copies the level statistics to host-mapped memory, zeroes the slots for the next pass, then stores `seq` at *h_seq
// (all pointers device-visible)
void launch_publish_stats(long long* lstats, const unsigned long long* l1max_bits, const double* within, int M,
                          long long* h_l, unsigned long long* h_l1, double* h_within, unsigned long long* h_seq,
                          unsigned long long seq, hipStream_t s);
bool has_cell_update(int NC);
// What k_cell_update publishes when `flags` is set (two words per cell: [0, M) "statistics out", [M, 2 M) "recursion
// done"; a cell's wave stores the low 32 bits of `seq` there, so nothing has to be reset; every pointer
// device-visible, the h_* ones host-mapped).  One extra workgroup, the last of the grid, does the publishing: it polls
// the first set of flags -- every cell raises its flag when its statistics are out, before the Levinson recursion,
// and carries on --, copies the level statistics (64 slots of 8 words, zeroed again afterwards), the L1 max of the
// codebook the pass ran on and the within-cell terms, then stores `seq` at *h_seq: the host spins on it, decides and
// launches the next pass while the kernel is still updating the cells.  The count of failed recursions follows when
// the second set of flags is up (*h_failed, then `seq` at *h_seq2).  Everything that crosses workgroups is a
// memory-side atomic or an agent-scope atomic store / load: no agent-scope fence (= L2 write-back) anywhere.
// (Measured alternatives: a ticket counter -- 256 returning atomics on one address -- costs ~25 us; __threadfence()
// per wave ~20 us at M = 1024.)
struct PublishArgs {
    unsigned int* flags;
    const unsigned long long* l1max_cur;
    long long* h_l;
    unsigned long long* h_l1;
    double* h_within;
    volatile unsigned long long* h_seq;
    long long* h_failed;
    volatile unsigned long long* h_seq2;
    unsigned long long seq;
    volatile unsigned long long* h_err;  // set to seq if a flag did not arrive within the publisher's spin cap
    const int* fb_count;                 // (optional) the fallback count of the pass's prefiltered sweep ...
    long long* h_fb;                     // ... and where the publisher copies it (-1 when there is none)
};
void launch_cell_update(const long long* rows, int M, int NC, const DevScalars* sc, const double* refl_in,
                        double* refl_out, double* cbq, double* cbm, unsigned long long* l1max_bits, double* within,
                        long long* lstats, hipStream_t s, bool zero_first = true, const int* ea = nullptr,
                        int* eC_biased = nullptr, const PublishArgs* pub = nullptr);
struct ZeroList {
    void* p[3];
    int words[3];  // 4-byte words
};
void launch_pass_prologue(long long* rows, int M, int NC, int what, const ZeroList& z, hipStream_t s);
void launch_finish_q(const long long* stats, int NC, DevScalars* sc, hipStream_t s);
void launch_init_codebook(const long long* stats, int NC, const DevScalars* sc, double* reflections, int* status,
                          hipStream_t s);
void launch_grow(const double* old_refl, int M, int NC, double* new_refl, hipStream_t s, const struct ZeroList* zero = nullptr);
void launch_codebook_prepare(const double* reflections, int M, int NC, double* cbq, unsigned long long* l1max_bits,
                             double* cbm, hipStream_t s);

}  // namespace e2vq
