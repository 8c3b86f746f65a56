// vq_io.h -- file helpers shared by the host driver and the CLI (internal)
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

int e2vq_set_error(const char* fmt, ...);

namespace e2vq_io {
int mkdirs_for(const char* path);
std::string basename_noext(const char* path);
// mirrors utl::resolve_filenames (/root/reference/src/utl/mod.rs:201-222): directories are
// walked recursively, only names ending in file_ext are kept, the result is sorted
int resolve_filenames(const std::vector<std::string>& given, const char* file_ext, std::vector<std::string>& out);
// mirrors utl::get_files_from_csv (src/utl/mod.rs:143-193): rows `tt,class,selection`
int files_from_csv(const std::string& csv, const std::string& tt, const std::string& class_name,
                   const std::string& subdir, const char* file_ext, const std::string* subdir_template,
                   std::vector<std::string>& out);
// frames [first, first + count) of a .prd file whose header has been validated by e2vq_prd_info
int prd_read_range(const char* path, int P, int64_t first, int64_t count, double* frames);
// ... by up to `threads` reader threads; *finite (optional) = no NaN / infinite value among them
int prd_read_range_mt(const char* path, int P, int64_t first, int64_t count, double* frames, int threads,
                      bool* finite = nullptr);
// a .seq written piecewise: header and final size, then symbol ranges (layout: src/sequence/mod.rs:49-75)
int seq_create(const char* path, const char* class_name, int M, int64_t T);
int seq_write_range(const char* path, int64_t t0, const uint16_t* sym, int64_t n);
// Levinson-Durbin on the host (src/lpc/lpca_r_rs.rs:8-43): status 0 / 1 (r0 == 0) / 2 (prediction error <= 0)
int lpca_r_host(int P, const double* r, double* rc, double* a);
int io_threads();  // reader threads per rank / worker
}  // namespace e2vq_io
