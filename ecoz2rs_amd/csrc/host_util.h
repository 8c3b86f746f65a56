// host_util.h -- host-side helpers shared by the `.seq` consumers (seq_models.cpp, hmm_host.cpp): Rust-style number
// formatting, small file helpers, and C12nResults, the classification report of /root/reference/src/c12n/mod.rs
// (which, per its line 5, is a translation of the C report the HMM classifier prints).  Internal.
#pragma once
#include "vq_io.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <string>
#include <vector>

namespace e2host {

inline const char* out_root()
{
    const char* v = getenv("ECOZ2_VQ_OUT_ROOT");
    return v && *v ? v : ".";
}

// `colored` 3.x colours only when stdout is a terminal (CLICOLOR / NO_COLOR / CLICOLOR_FORCE honoured)
inline bool use_colour()
{
    static const int on = [] {
        const char* force = getenv("CLICOLOR_FORCE");
        if (force && strcmp(force, "0") != 0) return 1;
        if (getenv("NO_COLOR")) return 0;
        const char* cc = getenv("CLICOLOR");
        if (cc && !strcmp(cc, "0")) return 0;
        return isatty(1) ? 1 : 0;
    }();
    return on != 0;
}

inline std::string coloured(const char* text, int code)
{
    if (!use_colour()) return text;
    char b[64];
    snprintf(b, sizeof b, "\x1b[%dm%s\x1b[0m", code, text);
    return b;
}

// ------------------------------------------------------------------------------------------
// Rust-style number formatting
// ------------------------------------------------------------------------------------------
// shortest round-trip digits and decimal exponent of a finite non-zero value: v = 0.d1d2... x 10^(exp10)
template <typename F>
void shortest_digits(F v, std::string& digits, int& point)
{
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v < 0 ? -v : v, std::chars_format::scientific);
    std::string s(buf, r.ptr);
    const size_t e = s.find('e');
    std::string mant = s.substr(0, e);
    const int ex = atoi(s.c_str() + e + 1);
    digits.clear();
    for (char c : mant)
        if (c != '.') digits.push_back(c);
    point = ex + 1;  // digits before the decimal point
}

// `{}` of f32 / f64 (Display): shortest digits, positional notation, "0" / "-0" for zeros, no trailing ".0" for
// integers?  -- Rust prints 1.0_f32 as "1", 0.5 as "0.5", 1e-7 as "0.0000001" (Display never uses an exponent)
template <typename F>
std::string rust_display(F v)
{
    if (v != v) return "NaN";
    if (v == (F)INFINITY) return "inf";
    if (v == -(F)INFINITY) return "-inf";
    if (v == 0) return std::signbit(v) ? "-0" : "0";
    std::string d;
    int point;
    shortest_digits(v, d, point);
    std::string out = v < 0 ? "-" : "";
    if (point <= 0) {
        out += "0.";
        out.append((size_t)(-point), '0');
        out += d;
    } else if ((size_t)point >= d.size()) {
        out += d;
        out.append((size_t)point - d.size(), '0');
    } else {
        out += d.substr(0, (size_t)point) + "." + d.substr((size_t)point);
    }
    return out;
}

// `{:e}` of f64 (LowerExp): shortest digits, d.ddde<exp> with no '+' and no padding
inline std::string rust_lower_exp(double v)
{
    if (v != v) return "NaN";
    if (v == INFINITY) return "inf";
    if (v == -INFINITY) return "-inf";
    if (v == 0) return std::signbit(v) ? "-0e0" : "0e0";
    std::string d;
    int point;
    shortest_digits(v, d, point);
    std::string out = v < 0 ? "-" : "";
    out += d[0];
    if (d.size() > 1) out += "." + d.substr(1);
    out += "e" + std::to_string(point - 1);
    return out;
}

// serde_json / ryu formatting of an f32 in [1e-5, 1e16): positional, always with a fractional part
inline std::string json_f32(float v)
{
    std::string s = rust_display(v);
    if (s.find('.') == std::string::npos && s.find('N') == std::string::npos && s.find('i') == std::string::npos) s += ".0";
    return s;
}

inline std::string json_string(const std::string& s)
{
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (c < 0x20) {
                    char b[8];
                    snprintf(b, sizeof b, "\\u%04x", c);
                    o += b;
                } else {
                    o.push_back((char)c);
                }
        }
    }
    return o + "\"";
}

// left-aligned text padded to `w` characters (`{:w$}` of a str), counting UTF-8 code points like Rust does
inline std::string pad_right(const std::string& s, size_t w)
{
    size_t n = 0;
    for (unsigned char c : s)
        if ((c & 0xC0) != 0x80) ++n;
    return n >= w ? s : s + std::string(w - n, ' ');
}

inline int read_file(const char* path, std::vector<unsigned char>& out)
{
    FILE* f = fopen(path, "rb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    unsigned char buf[1 << 16];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return 0;
}

inline int write_file(const std::string& path, const std::vector<unsigned char>& b)
{
    if (e2vq_io::mkdirs_for(path.c_str()) != 0) return e2vq_set_error("%s: cannot create directories", path.c_str());
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return e2vq_set_error("%s: %s", path.c_str(), strerror(errno));
    const bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
    if (fclose(f) != 0 || !ok) return e2vq_set_error("%s: write failed", path.c_str());
    return 0;
}

// the callers may be Python / Rust programs with their own stdout buffering: leave nothing in C stdio's buffer
struct FlushStdout {
    ~FlushStdout() { fflush(stdout); }
};

// ------------------------------------------------------------------------------------------
// c12n -- src/c12n/mod.rs
// ------------------------------------------------------------------------------------------
struct C12nResults {
    std::vector<std::string> model_class_names;
    std::vector<std::vector<int>> result, confusion;
    std::vector<std::string> y_true, y_pred;
    float last_accuracy = 0.f, last_avg_accuracy = 0.f;  // of the last report_results

    explicit C12nResults(std::vector<std::string> names) : model_class_names(std::move(names))
    {
        const size_t n = model_class_names.size();  // src/c12n/mod.rs:19-34
        result.assign(n + 1, std::vector<int>(n + 1, 0));
        confusion.assign(n + 1, std::vector<int>(n + 1, 0));
    }

    // src/c12n/mod.rs:36-104
    template <typename TitleFn>
    void add_case(size_t class_id, const std::string& seq_classname, const std::vector<double>& probs_in, bool show_ranked,
                  TitleFn title)
    {
        const size_t num_models = model_class_names.size();
        result[num_models][0] += 1;
        result[class_id][0] += 1;
        // sort given probabilities (ascending; slice::sort_by is a stable sort)
        std::vector<std::pair<size_t, double>> probs;
        for (size_t i = 0; i < probs_in.size(); ++i) probs.emplace_back(i, probs_in[i]);
        std::stable_sort(probs.begin(), probs.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
        const size_t predicted_id = probs[num_models - 1].first;
        const bool correct = class_id == predicted_id;
        fputs(correct ? coloured("*", 32).c_str() : coloured("_", 31).c_str(), stdout);
        fflush(stdout);
        y_true.push_back(seq_classname);
        y_pred.push_back(model_class_names[predicted_id]);
        if (show_ranked && !correct) {
            printf("%s\n", title().c_str());
            size_t index = 0;
            for (size_t r = num_models; r-- > 0; ++index) {
                const size_t model_id = probs[r].first;
                const std::string& model_class_name = model_class_names[r];  // (sic, :71)
                const char* mark = class_id == model_id ? "*" : "";
                printf("  [%2zu] %-1s model: <%2zu>  %s  : '%s'  r=%zu\n", index, mark, model_id,
                       rust_lower_exp(probs[model_id].second).c_str(), model_class_name.c_str(), r);  // (sic, :77)
                if (class_id == model_id) break;  // only show until corresponding model
            }
            printf("\n");
        }
        confusion[class_id][probs[num_models - 1].first] += 1;
        if (correct) {
            result[num_models][1] += 1;
            result[class_id][1] += 1;
        } else {
            for (size_t i = 1; i < num_models; ++i) {  // update order of recognized candidate
                if (probs[num_models - 1 - i].first == class_id) {
                    result[num_models][i + 1] += 1;
                    result[class_id][i + 1] += 1;
                    break;
                }
            }
        }
    }

    // src/c12n/mod.rs:106-223
    // c_report = true: the C original's report (hmm classify): the loop does reach the TOTAL row, and no JSON files
    int report_results(const std::vector<std::string>& class_names, const std::string& out_base_name,
                       bool c_report = false)
    {
        const size_t num_models = model_class_names.size();
        if (result[num_models][0] == 0) return 0;
        size_t margin = 0;
        for (size_t i = 0; i < class_names.size() && i < num_models; ++i)
            if (result[i][0] > 0) margin = std::max(margin, class_names[i].size());  // String::len(): bytes
        margin += 2;
        const std::string blank = pad_right("", margin);
        printf("\n\n");
        printf("%s Confusion matrix:\n", blank.c_str());
        printf("%s ", blank.c_str());
        printf("     ");
        for (size_t j = 0; j < num_models; ++j)
            if (result[j][0] > 0) printf("%3zu ", j);
        printf("    tests   errors\n");
        for (size_t i = 0; i < class_names.size() && i < num_models; ++i) {
            if (result[i][0] == 0) continue;
            printf("\n");
            printf("%s ", pad_right(class_names[i], margin).c_str());
            printf("%3zu  ", i);
            int num_errs = 0;  // in row
            for (size_t j = 0; j < num_models; ++j) {
                if (result[j][0] > 0) {
                    printf("%3d ", confusion[i][j]);
                    if (i != j) num_errs += confusion[i][j];
                }
            }
            printf("%8d%8d", result[i][0], num_errs);
        }
        printf("\n\n");
        printf("%s class     accuracy   tests       candidate order\n", blank.c_str());
        int num_classes = 0;
        float accuracy = 0.f, avg_accuracy = 0.f;
        // `.take(num_models + 1)` over the num_models names: the TOTAL branch (class_id == num_models) is never reached
        for (size_t class_id = 0; class_id < (c_report ? num_models + 1 : class_names.size()) && class_id < num_models + 1;
             ++class_id) {
            if (result[class_id][0] == 0) continue;
            const int num_tests = result[class_id][0], correct_tests = result[class_id][1];
            const float acc = (float)correct_tests / (float)num_tests;
            if (class_id < num_models) {
                num_classes += 1;
                avg_accuracy += acc;
                printf("%s ", pad_right(class_names[class_id], margin).c_str());
                printf("  %3zu    ", class_id);
            } else {
                printf("\n");
                printf("%s ", blank.c_str());
                printf("  TOTAL  ");
                accuracy = acc;
            }
            printf("  %6.2f%%    %4d       ", (double)(100.f * acc), num_tests);
            for (size_t i = 1; i <= num_models; ++i) printf("%4d ", result[class_id][i]);
            printf("\n");
        }
        accuracy *= 100.f;
        avg_accuracy = avg_accuracy * 100.f / (float)num_classes;
        printf("  avg_accuracy  %6.2f%%\n", (double)avg_accuracy);
        printf("\n");
        last_accuracy = accuracy;
        last_avg_accuracy = avg_accuracy;
        if (c_report) return 0;
        // utl::save_json = serde_json::to_writer_pretty (src/utl/mod.rs:270-275)
        const std::string dir = std::string(out_root()) + "/";
        const std::string out_summary = out_base_name + "_classification.json";
        {
            std::string j = "{\n  \"accuracy\": " + json_f32(accuracy) + ",\n  \"avg_accuracy\": " + json_f32(avg_accuracy) + "\n}";
            if (write_file(dir + out_summary, std::vector<unsigned char>(j.begin(), j.end()))) return 1;
        }
        printf("%s saved\n", out_summary.c_str());
        const std::string out_true_pred = out_base_name + "_y_true_pred.json";
        {
            auto arr = [](const std::vector<std::string>& v) {
                if (v.empty()) return std::string("[]");
                std::string s = "[\n";
                for (size_t i = 0; i < v.size(); ++i) s += "    " + json_string(v[i]) + (i + 1 < v.size() ? ",\n" : "\n");
                return s + "  ]";
            };
            std::string j = "{\n  \"y_true\": " + arr(y_true) + ",\n  \"y_pred\": " + arr(y_pred) + "\n}";
            if (write_file(dir + out_true_pred, std::vector<unsigned char>(j.begin(), j.end()))) return 1;
        }
        printf("%s saved\n", out_true_pred.c_str());
        return 0;
    }
};

}  // namespace e2host
