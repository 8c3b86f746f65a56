// vq_io.cpp -- .prd / .cbook / .seq files, file-list resolution, synthetic predictor frames.
//
// Formats.  Every file starts with a 16-byte NUL-padded ident and a
// 96-byte NUL-padded class name (FILE_IDENT_LEN / MAX_CLASS_NAME_LEN,
// /root/reference/src/utl/mod.rs:19-20), then little-endian fields:
//   .prd    "<predictor>"  u32 T, u32 P, T*(P+1) f64 (gain-normalised autocorrelations)
//   .cbook  "<codebook>"   u32 P, u32 M, M*(P+1) f64 (reflection coefficients, [0] unused)
//   .seq    "<sequence>"   u32 T, u32 M, T * u16 symbols      (src/sequence/mod.rs:49-75)
#include "../../include/ecoz2_vq.h"
#include "vq_io.h"

#include <dirent.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <fstream>
#include <sstream>
#include <thread>
#include <vector>

namespace e2vq_io {

int mkdirs_for(const char* path)
{
    std::string p(path);
    for (size_t i = 1; i < p.size(); ++i) {
        if (p[i] == '/') {
            p[i] = 0;
            if (mkdir(p.c_str(), 0777) != 0 && errno != EEXIST) return -1;
            p[i] = '/';
        }
    }
    return 0;
}

std::string basename_noext(const char* path)
{
    std::string p(path);
    const size_t slash = p.find_last_of('/');
    if (slash != std::string::npos) p = p.substr(slash + 1);
    const size_t dot = p.find_last_of('.');
    if (dot != std::string::npos && dot > 0) p = p.substr(0, dot);
    return p;
}

static bool ends_with(const std::string& s, const char* ext)
{
    const size_t n = strlen(ext);
    return s.size() >= n && s.compare(s.size() - n, n, ext) == 0;
}

static void walk(const std::string& dir, const char* ext, std::vector<std::string>& out)
{
    DIR* d = opendir(dir.c_str());
    if (!d) return;
    while (struct dirent* e = readdir(d)) {
        if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
        const std::string p = dir + "/" + e->d_name;
        struct stat st;
        if (stat(p.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode))
            walk(p, ext, out);
        else if (S_ISREG(st.st_mode) && ends_with(p, ext))
            out.push_back(p);
    }
    closedir(d);
}

int resolve_filenames(const std::vector<std::string>& given, const char* file_ext, std::vector<std::string>& out)
{
    out.clear();
    for (const std::string& g : given) {
        struct stat st;
        if (stat(g.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) {
            std::string dir = g;
            while (dir.size() > 1 && dir.back() == '/') dir.pop_back();
            walk(dir, file_ext, out);
        } else if (S_ISREG(st.st_mode) && ends_with(g, file_ext)) {
            out.push_back(g);
        }
    }
    // `filenames.sort()` on Vec<PathBuf> (src/utl/mod.rs:216-218): paths compare component by component, not
    // byte by byte ("a/b.prd" sorts before "a-b/c.prd" although '-' < '/')
    std::sort(out.begin(), out.end(), [](const std::string& a, const std::string& b) {
        size_t i = 0, j = 0;
        for (;;) {
            while (i < a.size() && a[i] == '/') ++i;  // (repeated separators do not make components)
            while (j < b.size() && b[j] == '/') ++j;
            if (i >= a.size() || j >= b.size()) return i >= a.size() && j < b.size();
            const size_t ie = std::min(a.find('/', i), a.size()), je = std::min(b.find('/', j), b.size());
            const int c = a.compare(i, ie - i, b, j, je - j);
            if (c != 0) return c < 0;
            i = ie;
            j = je;
        }
    });
    return 0;
}

static std::string replace_all(std::string s, const std::string& what, const std::string& with)
{
    for (size_t pos = 0; (pos = s.find(what, pos)) != std::string::npos; pos += with.size())
        s.replace(pos, what.size(), with);
    return s;
}

int files_from_csv(const std::string& csv, const std::string& tt, const std::string& class_name,
                   const std::string& subdir, const char* file_ext, const std::string* subdir_template,
                   std::vector<std::string>& out)
{
    std::ifstream in(csv);
    if (!in) return e2vq_set_error("%s: cannot open", csv.c_str());
    out.clear();
    std::string line;
    bool header = true;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        if (header) {  // first record is the header `tt,class,selection`
            header = false;
            continue;
        }
        std::stringstream ss(line);
        std::string rtt, rclass, rsel;
        if (!std::getline(ss, rtt, ',') || !std::getline(ss, rclass, ',') || !std::getline(ss, rsel, ',')) continue;
        if (tt != rtt) continue;
        if (!class_name.empty() && class_name != rclass) continue;
        if (subdir_template)
            out.push_back(replace_all(replace_all(*subdir_template, "{class}", rclass), "{selection}", rsel));
        else
            out.push_back("data/" + subdir + "/" + rclass + "/" + rsel + file_ext);
    }
    if (out.empty()) return e2vq_set_error("No %s given in given file", subdir.c_str());
    return 0;
}

}  // namespace e2vq_io

// ------------------------------------------------------------------------------------------
namespace {

void put_u32(FILE* f, uint32_t v)
{
    const unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16),
                                (unsigned char)(v >> 24)};
    fwrite(b, 1, 4, f);
}

bool get_u32(FILE* f, uint32_t* v)
{
    unsigned char b[4];
    if (fread(b, 1, 4, f) != 4) return false;
    *v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    return true;
}

void put_header(FILE* f, const char* ident, const char* class_name)
{
    char id[16] = {0}, cn[96] = {0};
    strncpy(id, ident, sizeof id - 1);
    strncpy(cn, class_name, sizeof cn - 1);
    fwrite(id, 1, sizeof id, f);
    fwrite(cn, 1, sizeof cn, f);
}

// 0 ok; message set otherwise
int get_header(FILE* f, const char* path, const char* ident, const char* what, char class_name[96])
{
    char id[16];
    if (fread(id, 1, sizeof id, f) != sizeof id) return e2vq_set_error("%s: truncated header", path);
    if (strncmp(id, ident, strlen(ident)) != 0) return e2vq_set_error("%s: Not a %s", path, what);
    if (fread(class_name, 1, 96, f) != 96) return e2vq_set_error("%s: truncated header", path);
    class_name[95] = 0;
    return 0;
}

// The order of the two u32 header fields of .prd / .cbook is not pinned by anything in the reference (the C
// reader/writer is absent; only .seq is pinned).  This repo writes (count, order) for .prd and (order, count) for
// .cbook; on reading, the payload size decides: 120 + count*(order+1)*8 must equal the file size, and if only the
// swapped reading satisfies it the fields are taken swapped.  Anything else is rejected.
int resolve_fields(FILE* f, const char* path, uint32_t a, uint32_t b, bool a_is_count, uint32_t* count, uint32_t* order)
{
    struct stat st;
    if (fstat(fileno(f), &st) != 0) return e2vq_set_error("%s: cannot stat", path);
    const unsigned long long payload = (unsigned long long)st.st_size - 120ull;
    const uint32_t c0 = a_is_count ? a : b, o0 = a_is_count ? b : a;
    if ((unsigned long long)c0 * (o0 + 1ull) * 8ull == payload && o0 >= 1 && o0 <= 200) {
        *count = c0;
        *order = o0;
        return 0;
    }
    if ((unsigned long long)o0 * (c0 + 1ull) * 8ull == payload && c0 >= 1 && c0 <= 200) {
        *count = o0;
        *order = c0;
        return 0;
    }
    return e2vq_set_error("%s: header fields (%u, %u) do not match the payload size %llu", path, a, b, payload);
}

}  // namespace

extern "C" int e2vq_prd_info(const char* path, char class_name[96], int* P, int64_t* T)
{
    FILE* f = fopen(path, "rb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    uint32_t t = 0, p = 0;
    int rc = get_header(f, path, "<predictor>", "predictor", class_name);
    if (!rc && (!get_u32(f, &t) || !get_u32(f, &p))) rc = e2vq_set_error("%s: truncated header", path);
    if (!rc) rc = resolve_fields(f, path, t, p, true, &t, &p);
    fclose(f);
    if (rc) return rc;
    *P = (int)p;
    *T = (int64_t)t;
    return 0;
}

extern "C" int e2vq_prd_read(const char* path, double* frames, int64_t capacity_frames)
{
    FILE* f = fopen(path, "rb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    char cls[96];
    uint32_t t = 0, p = 0;
    int rc = get_header(f, path, "<predictor>", "predictor", cls);
    if (!rc && (!get_u32(f, &t) || !get_u32(f, &p))) rc = e2vq_set_error("%s: truncated header", path);
    if (!rc) rc = resolve_fields(f, path, t, p, true, &t, &p);
    if (!rc && (int64_t)t > capacity_frames) rc = e2vq_set_error("%s: %u vectors exceed the buffer", path, t);
    if (!rc && fread(frames, sizeof(double), (size_t)t * (p + 1), f) != (size_t)t * (p + 1))
        rc = e2vq_set_error("%s: truncated payload", path);
    fclose(f);
    return rc;
}

// frames [first, first + count) of a .prd whose header e2vq_prd_info has validated (payload at byte 120)
int e2vq_io::prd_read_range(const char* path, int P, int64_t first, int64_t count, double* frames)
{
    FILE* f = fopen(path, "rb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    const size_t row = (size_t)(P + 1) * sizeof(double);
    int rc = 0;
    if (fseeko(f, (off_t)(120 + (unsigned long long)first * row), SEEK_SET) != 0) rc = e2vq_set_error("%s: seek failed", path);
    if (!rc && fread(frames, row, (size_t)count, f) != (size_t)count) rc = e2vq_set_error("%s: truncated payload", path);
    fclose(f);
    return rc;
}

// the same range read by up to `threads` threads (page-cache reads run at ~10 GB/s per thread, PCIe at ~50 GB/s: one
// reader thread per GPU starves the upload).  finite (optional): set to false if any value read is NaN / infinite.
int e2vq_io::prd_read_range_mt(const char* path, int P, int64_t first, int64_t count, double* frames, int threads,
                               bool* finite)
{
    const size_t row = (size_t)(P + 1) * sizeof(double);
    const int64_t min_rows = std::max<int64_t>(1, (int64_t)((8u << 20) / row));  // at least 8 MB per thread
    int n = (int)std::min<int64_t>(std::max(1, threads), (count + min_rows - 1) / min_rows);
    if (n < 1) n = 1;
    std::vector<int> rcs((size_t)n, 0);
    std::vector<char> fin((size_t)n, 1);
    std::vector<std::string> errs((size_t)n);
    auto piece = [&](int k) {
        const int64_t a = count * k / n, b = count * (k + 1) / n;
        if (b <= a) return;
        rcs[(size_t)k] = prd_read_range(path, P, first + a, b - a, frames + (size_t)a * (P + 1));
        if (rcs[(size_t)k]) {
            errs[(size_t)k] = e2vq_last_error();  // (thread-local: handed to the calling thread below)
            return;
        }
        if (finite) {
            const double* v = frames + (size_t)a * (P + 1);
            const size_t m = (size_t)(b - a) * (P + 1);
            bool ok = true;
            for (size_t i = 0; i < m; ++i) ok &= std::isfinite(v[i]);
            fin[(size_t)k] = ok;
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n; ++k) th.emplace_back(piece, k);
    piece(0);
    for (auto& t : th) t.join();
    for (int k = 0; k < n; ++k)
        if (rcs[(size_t)k]) return e2vq_set_error("%s", errs[(size_t)k].empty() ? "read failed" : errs[(size_t)k].c_str());
    if (finite) {
        *finite = true;
        for (int k = 0; k < n; ++k) *finite = *finite && fin[(size_t)k];
    }
    return 0;
}

int e2vq_io::io_threads()
{
    return 4;
}

extern "C" int e2vq_prd_write(const char* path, const char* class_name, int P, const double* frames, int64_t T)
{
    // the header carries the vector count as u32 (src/sequence/mod.rs:60 for .seq; .prd by analogy)
    if (T < 0 || T > (int64_t)UINT32_MAX) return e2vq_set_error("%s: %lld vectors do not fit the u32 header field", path, (long long)T);
    if (P < 1) return e2vq_set_error("%s: prediction order %d", path, P);
    if (e2vq_io::mkdirs_for(path) != 0) return e2vq_set_error("%s: cannot create directories", path);
    FILE* f = fopen(path, "wb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    put_header(f, "<predictor>", class_name);
    put_u32(f, (uint32_t)T);
    put_u32(f, (uint32_t)P);
    const size_t n = (size_t)T * (P + 1);
    const bool ok = fwrite(frames, sizeof(double), n, f) == n;
    if (fclose(f) != 0 || !ok) return e2vq_set_error("%s: write failed", path);
    return 0;
}

extern "C" int e2vq_cbook_info(const char* path, char class_name[96], int* P, int* M)
{
    FILE* f = fopen(path, "rb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    uint32_t p = 0, m = 0;
    int rc = get_header(f, path, "<codebook>", "codebook", class_name);
    if (!rc && (!get_u32(f, &p) || !get_u32(f, &m))) rc = e2vq_set_error("%s: truncated header", path);
    if (!rc) rc = resolve_fields(f, path, p, m, false, &m, &p);
    fclose(f);
    if (rc) return rc;
    if (p < 1 || p > 200 || m < 1 || m > 65536) return e2vq_set_error("%s: implausible P=%u M=%u", path, p, m);
    *P = (int)p;
    *M = (int)m;
    return 0;
}

extern "C" int e2vq_cbook_read(const char* path, double* reflections, int capacity_codewords)
{
    FILE* f = fopen(path, "rb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    char cls[96];
    uint32_t p = 0, m = 0;
    int rc = get_header(f, path, "<codebook>", "codebook", cls);
    if (!rc && (!get_u32(f, &p) || !get_u32(f, &m))) rc = e2vq_set_error("%s: truncated header", path);
    if (!rc) rc = resolve_fields(f, path, p, m, false, &m, &p);
    if (!rc && (int)m > capacity_codewords) rc = e2vq_set_error("%s: %u codewords exceed the buffer", path, m);
    if (!rc && fread(reflections, sizeof(double), (size_t)m * (p + 1), f) != (size_t)m * (p + 1))
        rc = e2vq_set_error("%s: truncated payload", path);
    fclose(f);
    return rc;
}

extern "C" int e2vq_cbook_write(const char* path, const char* class_name, int P, int M, const double* reflections)
{
    if (e2vq_io::mkdirs_for(path) != 0) return e2vq_set_error("%s: cannot create directories", path);
    FILE* f = fopen(path, "wb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    put_header(f, "<codebook>", class_name);
    put_u32(f, (uint32_t)P);
    put_u32(f, (uint32_t)M);
    const size_t n = (size_t)M * (P + 1);
    const bool ok = fwrite(reflections, sizeof(double), n, f) == n;
    if (fclose(f) != 0 || !ok) return e2vq_set_error("%s: write failed", path);
    return 0;
}

extern "C" int e2vq_seq_write(const char* path, const char* class_name, int M, const uint16_t* sym, int64_t T)
{
    if (T < 0 || T > (int64_t)UINT32_MAX) return e2vq_set_error("%s: %lld symbols do not fit the u32 header field", path, (long long)T);
    if (M < 1 || M > 65536) return e2vq_set_error("%s: codebook size %d does not fit u16 symbols", path, M);
    if (e2vq_io::mkdirs_for(path) != 0) return e2vq_set_error("%s: cannot create directories", path);
    FILE* f = fopen(path, "wb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    put_header(f, "<sequence>", class_name);
    put_u32(f, (uint32_t)T);
    put_u32(f, (uint32_t)M);
    std::vector<unsigned char> buf((size_t)T * 2);
    for (int64_t t = 0; t < T; ++t) {
        buf[(size_t)2 * t] = (unsigned char)sym[t];
        buf[(size_t)2 * t + 1] = (unsigned char)(sym[t] >> 8);
    }
    const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    if (fclose(f) != 0 || !ok) return e2vq_set_error("%s: write failed", path);
    return 0;
}

// a .seq written piecewise (a .prd larger than a quantize chunk is split over workers): header + final size first, then
// each worker stores its symbol range at byte 120 + 2 t0
int e2vq_io::seq_create(const char* path, const char* class_name, int M, int64_t T)
{
    if (T < 0 || T > (int64_t)UINT32_MAX) return e2vq_set_error("%s: %lld symbols do not fit the u32 header field", path, (long long)T);
    if (M < 1 || M > 65536) return e2vq_set_error("%s: codebook size %d does not fit u16 symbols", path, M);
    if (e2vq_io::mkdirs_for(path) != 0) return e2vq_set_error("%s: cannot create directories", path);
    FILE* f = fopen(path, "wb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    put_header(f, "<sequence>", class_name);
    put_u32(f, (uint32_t)T);
    put_u32(f, (uint32_t)M);
    bool ok = fflush(f) == 0 && ftruncate(fileno(f), (off_t)(120 + 2 * T)) == 0;
    if (fclose(f) != 0 || !ok) return e2vq_set_error("%s: write failed", path);
    return 0;
}

int e2vq_io::seq_write_range(const char* path, int64_t t0, const uint16_t* sym, int64_t n)
{
    FILE* f = fopen(path, "r+b");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    std::vector<unsigned char> buf((size_t)n * 2);
    for (int64_t t = 0; t < n; ++t) {
        buf[(size_t)2 * t] = (unsigned char)sym[t];
        buf[(size_t)2 * t + 1] = (unsigned char)(sym[t] >> 8);
    }
    bool ok = fseeko(f, (off_t)(120 + 2 * t0), SEEK_SET) == 0 && fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    if (fclose(f) != 0 || !ok) return e2vq_set_error("%s: write failed", path);
    return 0;
}

// ------------------------------------------------------------------------------------------
// synthetic predictor frames: valid gain-normalised autocorrelation vectors.
// class j prototype reflections k_j[i] ~ U(-0.7,0.7)*0.97^i; frame: k = clamp(k_j + noise, +-0.95),
// noise ~ N(0, 0.05^2) approximated by a 12-uniform sum (pure arithmetic, no libm);
// step-up k -> a, inverse Levinson with r0 = 1, then divide by E_P (src/lpc/lpc_rs.rs:125-131).
// Counter based: frame t depends only on (seed, n_classes, P, t).
// ------------------------------------------------------------------------------------------
// ---- prd show ---------------------------------------------------------------------------------------------------
// replaces `fn ecoz2_prd_show_file(prd_filename, show_reflections, from, to)` (src/ecoz2_lib/mod.rs:89-94; caller
// src/prd/mod.rs:99, options :30-62: from defaults to 1, to = 0 means "up to P").  Output as in notes.md:77-85: a
// `# <file>:` line, `# className='..', T=.., P=..`, the column names r<from>..r<to> (k<..> for reflections), then one
// line of %.5f values per vector.  Reflections come from the Levinson recursion on the vector's autocorrelation,
// restated from src/lpc/lpca_r_rs.rs:8-43 (status 1: r0 == 0, 2: prediction error <= 0; such rows print zeros).
int e2vq_io::lpca_r_host(int P, const double* r, double* rc, double* a)
{
    const double r0 = r[0];
    if (0.0 == r0) return 1;
    double pe = r0;
    a[0] = 1.0;
    for (int k = 1; k <= P; ++k) {
        double sum = 0.0;
        for (int i = 1; i <= k; ++i) sum -= a[k - i] * r[i];
        const double akk = sum / pe;
        rc[k] = akk;
        a[k] = akk;
        for (int i = 1; i <= (k >> 1); ++i) {
            const double ai = a[i], aj = a[k - i];
            a[i] = ai + akk * aj;
            a[k - i] = aj + akk * ai;
        }
        pe *= 1.0 - akk * akk;
        if (pe <= 0.0) return 2;
    }
    return 0;
}

extern "C" int ecoz2_prd_show_file(const char* prd_filename, int show_reflections, int from, int to)
{
    if (!prd_filename) return e2vq_set_error("ecoz2_prd_show_file: bad arguments");
    char cls[96];
    int P;
    int64_t T;
    if (e2vq_prd_info(prd_filename, cls, &P, &T)) return 1;
    if (to <= 0 || to > P) to = P;
    if (from < 0) from = 0;
    printf("# %s:\n# className='%s', T=%lld, P=%d\n", prd_filename, cls, (long long)T, P);
    const char* name = show_reflections ? "k" : "r";
    for (int n = from; n <= to; ++n) printf("%s%s%d", n == from ? "" : ",", name, n);
    printf("\n");
    const int NC = P + 1;
    const int64_t CH = 4096;  // vectors per read
    std::vector<double> fr((size_t)std::min<int64_t>(CH, std::max<int64_t>(T, 1)) * NC), rc((size_t)NC), a((size_t)NC);
    for (int64_t t0 = 0; t0 < T; t0 += CH) {
        const int64_t n = std::min(CH, T - t0);
        if (e2vq_io::prd_read_range(prd_filename, P, t0, n, fr.data())) return 1;
        for (int64_t t = 0; t < n; ++t) {
            const double* v = fr.data() + (size_t)t * NC;
            if (show_reflections) {
                std::fill(rc.begin(), rc.end(), 0.0);
                if (e2vq_io::lpca_r_host(P, v, rc.data(), a.data()) != 0) std::fill(rc.begin(), rc.end(), 0.0);
                v = rc.data();
            }
            for (int k = from; k <= to; ++k) printf("%s%.5f", k == from ? "" : ",", v[k]);
            printf("\n");
        }
    }
    fflush(stdout);
    return 0;
}

namespace {

inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

// kind 1: the mean reflections of the continuum -- orders 1 ... 10 from the Levinson recursion on the first vector the
// reference prints of its whale-song predictor file (notes.md:80-85: r = 2.35985, 0.32431, -0.69968, ...), beyond them the
// same alternating pattern fading out, so that r[0] = 1 / E comes out at 2-3 as in that file
inline double continuum_mean(int i)
{
    static const double k10[10] = {-0.1374, 0.3215, -0.1127, 0.2911, -0.1795, 0.2471, -0.1894, 0.2684, -0.1666, 0.2325};
    if (i <= 10) return k10[i - 1];
    return ((i & 1) ? -0.13 : 0.20) * pow(0.96, (double)(i - 10));
}

void synth_one(uint64_t seed, int kind, int n_classes, double noise, int P, int64_t t, double* r)
{
    const uint64_t ft = splitmix64(seed ^ splitmix64((uint64_t)t + 0x51ED270Bull));
    const int cls = (int)(ft % (uint64_t)n_classes);
    double k[201], a[201];
    double decay = 1.0;
    for (int i = 1; i <= P; ++i) {
        decay *= 0.97;
        double proto;
        if (kind == 1) {
            // a smooth path through reflection space: n_classes slow sinusoids per coefficient (periods 50 ... 5000 frames,
            // i.e. 0.75 ... 75 s of 15 ms offsets), amplitudes fading with the order
            proto = continuum_mean(i);
            const double amp = 0.13 * (i <= 10 ? 1.0 : pow(0.96, (double)(i - 10))) / sqrt((double)n_classes);
            for (int j = 0; j < n_classes; ++j) {
                const uint64_t hs = splitmix64(seed * 0x2545F4914F6CDD1Dull + (uint64_t)j * 1000003ull + (uint64_t)i);
                const double period = 50.0 * pow(100.0, u01(splitmix64(hs ^ 0xA5A5A5A5ull)));
                proto += amp * sin(6.283185307179586 * ((double)t / period + u01(hs)));
            }
        } else {
            const uint64_t hp = splitmix64(seed * 0x2545F4914F6CDD1Dull + (uint64_t)cls * 1000003ull + (uint64_t)i);
            proto = (u01(hp) * 1.4 - 0.7) * decay;
        }
        double g = -6.0;
        uint64_t h = splitmix64(ft + (uint64_t)i * 0xD1B54A32D192ED03ull);
        for (int j = 0; j < 12; ++j) {
            h = splitmix64(h);
            g += u01(h);
        }
        double v = proto + noise * g;
        if (v > 0.95) v = 0.95;
        if (v < -0.95) v = -0.95;
        k[i] = v;
    }
    // inverse Levinson: r[0] = 1, E = 1
    double E = 1.0;
    r[0] = 1.0;
    a[0] = 1.0;
    for (int kk = 1; kk <= P; ++kk) {
        const double akk = k[kk];
        double s = 0.0;
        for (int i = 1; i < kk; ++i) s += a[kk - i] * r[i];
        r[kk] = -akk * E - s;
        a[kk] = akk;
        for (int i = 1; i <= (kk >> 1); ++i) {
            const double ai = a[i], aj = a[kk - i];
            a[i] = ai + akk * aj;
            a[kk - i] = aj + akk * ai;
        }
        E *= 1.0 - akk * akk;
    }
    for (int n = 0; n <= P; ++n) r[n] /= E;
}

}  // namespace

extern "C" int e2vq_synth_frames_kind(uint64_t seed, int kind, int n_classes, double noise, int P, int64_t first, int64_t count,
                                      double* frames)
{
    if (n_classes < 1 || P < 1 || P > 200 || count < 0 || kind < 0 || kind > 1 || !(noise >= 0.0 && noise <= 1.0))
        return e2vq_set_error("e2vq_synth_frames: bad arguments");
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if ((int64_t)nt > count / 1024 + 1) nt = (unsigned)(count / 1024 + 1);
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nt; ++w) {
        th.emplace_back([=]() {
            const int64_t a = count * w / nt, b = count * (w + 1) / nt;
            for (int64_t i = a; i < b; ++i) synth_one(seed, kind, n_classes, noise, P, first + i, frames + (size_t)i * (P + 1));
        });
    }
    for (auto& t : th) t.join();
    return 0;
}

extern "C" int e2vq_synth_frames(uint64_t seed, int n_classes, int P, int64_t first, int64_t count, double* frames)
{
    return e2vq_synth_frames_kind(seed, 0, n_classes, 0.05, P, first, count, frames);
}
