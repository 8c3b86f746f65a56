// seq_models.cpp -- consumers of the `.seq` files the VQ path emits:
//   nb   naive-Bayes classifier            /root/reference/src/nb/nbayes.rs, src/nb/mod.rs
//   mm   first-order Markov-model classifier /root/reference/src/mm/markov.rs, src/mm/mod.rs
//   c12n classification report             /root/reference/src/c12n/mod.rs
// These are pure Rust (host, CPU) in the reference and are restated here as host C++ with the reference's exact
// arithmetic: f64 + log10 for nb, f32 + log10f for mm, a stable ascending sort for the ranking.  Model files are the
// CBOR documents `utl::save_ser` writes (serde_cbor 0.11.2 of the derive(Serialize) structs; ndarray 0.17 arrays as
// {"v":1,"dim":[..],"data":[..]}); the reader accepts any well-formed CBOR encoding of the same documents.
// Known quirks of the reference are kept on purpose (drop-in output): report_results never reaches its TOTAL row
// (`take(num_models + 1)` over num_models names, src/c12n/mod.rs:170), so "accuracy" in the JSON summary stays 0;
// the ranked listing indexes names by rank and probabilities by model id (src/c12n/mod.rs:71-79).
#include "../../include/ecoz2_classify.h"
#include "../../include/ecoz2_vq.h"
#include "host_util.h"
#include "vq_io.h"

#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace e2host;

namespace {

// ------------------------------------------------------------------------------------------
// CBOR (RFC 8949) -- writer in serde_cbor 0.11's encoding, reader for any well-formed encoding
// ------------------------------------------------------------------------------------------
struct CborOut {
    std::vector<unsigned char> b;
    void head(int major, uint64_t v)
    {
        const unsigned char m = (unsigned char)(major << 5);
        if (v < 24) b.push_back(m | (unsigned char)v);
        else if (v <= 0xff) { b.push_back(m | 24); b.push_back((unsigned char)v); }
        else if (v <= 0xffff) { b.push_back(m | 25); be(v, 2); }
        else if (v <= 0xffffffffull) { b.push_back(m | 26); be(v, 4); }
        else { b.push_back(m | 27); be(v, 8); }
    }
    void be(uint64_t v, int n)
    {
        for (int i = n - 1; i >= 0; --i) b.push_back((unsigned char)(v >> (8 * i)));
    }
    void uint(uint64_t v) { head(0, v); }
    void text(const std::string& s)
    {
        head(3, s.size());
        b.insert(b.end(), s.begin(), s.end());
    }
    void array(uint64_t n) { head(4, n); }
    void map(uint64_t n) { head(5, n); }
    // serde_cbor serialize_f32: the half-precision form when it is lossless, else the 4-byte form
    void f32(float v)
    {
        uint32_t u;
        memcpy(&u, &v, 4);
        if (v != v) { b.push_back(0xf9); b.push_back(0x7e); b.push_back(0x00); return; }
        if (isinf(v)) { b.push_back(0xf9); b.push_back(v > 0 ? 0x7c : 0xfc); b.push_back(0x00); return; }
        uint16_t h;
        if (to_half_exact(u, &h)) { b.push_back(0xf9); be(h, 2); return; }
        b.push_back(0xfa);
        be(u, 4);
    }
    // true when the f32 bit pattern is exactly representable in IEEE half precision
    static bool to_half_exact(uint32_t u, uint16_t* h)
    {
        const uint32_t sign = (u >> 16) & 0x8000u, man = u & 0x7fffffu;
        const int e = (int)((u >> 23) & 0xff);
        if (e == 0) {  // f32 zero / subnormal: only zero fits
            if (man != 0) return false;
            *h = (uint16_t)sign;
            return true;
        }
        const int he = e - 127 + 15;
        if (he >= 31) return false;
        if (he >= 1) {  // normal half: the low 13 mantissa bits must be zero
            if (man & 0x1fffu) return false;
            *h = (uint16_t)(sign | ((uint32_t)he << 10) | (man >> 13));
            return true;
        }
        // subnormal half: value = (1.man) x 2^(e-127) must be a multiple of 2^-24
        const int shift = 14 - he;  // bits dropped from the 24-bit significand (he <= 0 -> shift >= 14)
        if (shift > 24) return false;
        const uint32_t sig = man | 0x800000u;
        if (shift == 24 ? true : (sig & ((1u << shift) - 1)) != 0) return false;
        *h = (uint16_t)(sign | (sig >> shift));
        return true;
    }
};

struct CborVal {
    enum Kind { UINT, NINT, FLOAT, TEXT, BYTES, ARRAY, MAP, SIMPLE } kind = SIMPLE;
    uint64_t u = 0;  // UINT value; NINT: -1 - u
    double f = 0.0;
    std::string s;
    std::vector<CborVal> items;                           // ARRAY
    std::vector<std::pair<std::string, CborVal>> fields;  // MAP with text keys
    const CborVal* get(const char* key) const
    {
        for (const auto& kv : fields)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool number(double* out) const
    {
        if (kind == UINT) { *out = (double)u; return true; }
        if (kind == NINT) { *out = -1.0 - (double)u; return true; }
        if (kind == FLOAT) { *out = f; return true; }
        return false;
    }
};

struct CborIn {
    const unsigned char* p;
    const unsigned char* end;
    bool ok = true;
    int depth = 0;
    uint64_t be(int n)
    {
        if (end - p < n) { ok = false; return 0; }
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 8) | *p++;
        return v;
    }
    static double half_to_double(uint16_t h)
    {
        const int e = (h >> 10) & 31, m = h & 1023;
        double v = e == 0 ? ldexp((double)m, -24) : e == 31 ? (m ? NAN : INFINITY) : ldexp((double)(m + 1024), e - 25);
        return (h & 0x8000) ? -v : v;
    }
    bool value(CborVal& out)
    {
        if (!ok || p >= end || ++depth > 64) return ok = false;
        const unsigned char ib = *p++;
        const int major = ib >> 5, ai = ib & 31;
        uint64_t arg = 0;
        bool indefinite = false;
        if (ai < 24) arg = (uint64_t)ai;
        else if (ai == 24) arg = be(1);
        else if (ai == 25) arg = be(2);
        else if (ai == 26) arg = be(4);
        else if (ai == 27) arg = be(8);
        else if (ai == 31 && (major >= 2 && major <= 5)) indefinite = true;
        else if (!(major == 7 && ai == 31)) return ok = false;
        if (!ok) return false;
        switch (major) {
            case 0: out.kind = CborVal::UINT; out.u = arg; break;
            case 1: out.kind = CborVal::NINT; out.u = arg; break;
            case 2:
            case 3: {
                out.kind = major == 2 ? CborVal::BYTES : CborVal::TEXT;
                if (indefinite) {
                    while (ok && p < end && *p != 0xff) {
                        CborVal chunk;
                        if (!value(chunk) || chunk.kind != out.kind) return ok = false;
                        out.s += chunk.s;
                    }
                    if (p >= end) return ok = false;
                    ++p;
                } else {
                    if ((uint64_t)(end - p) < arg) return ok = false;
                    out.s.assign((const char*)p, (size_t)arg);
                    p += arg;
                }
                break;
            }
            case 4: {
                out.kind = CborVal::ARRAY;
                if (!indefinite && arg > (uint64_t)(end - p)) return ok = false;
                if (!indefinite) out.items.reserve((size_t)arg);
                for (uint64_t i = 0; indefinite ? (p < end && *p != 0xff) : i < arg; ++i) {
                    out.items.emplace_back();
                    if (!value(out.items.back())) return false;
                }
                if (indefinite) {
                    if (p >= end) return ok = false;
                    ++p;
                }
                break;
            }
            case 5: {
                out.kind = CborVal::MAP;
                if (!indefinite && arg > (uint64_t)(end - p)) return ok = false;
                for (uint64_t i = 0; indefinite ? (p < end && *p != 0xff) : i < arg; ++i) {
                    CborVal k, v;
                    if (!value(k) || !value(v)) return false;
                    if (k.kind != CborVal::TEXT) return ok = false;
                    out.fields.emplace_back(std::move(k.s), std::move(v));
                }
                if (indefinite) {
                    if (p >= end) return ok = false;
                    ++p;
                }
                break;
            }
            case 6: --depth; return value(out);  // tag: ignored
            default: {
                if (ai == 25) { out.kind = CborVal::FLOAT; out.f = half_to_double((uint16_t)arg); }
                else if (ai == 26) { uint32_t u = (uint32_t)arg; float f; memcpy(&f, &u, 4); out.kind = CborVal::FLOAT; out.f = f; }
                else if (ai == 27) { double d; memcpy(&d, &arg, 8); out.kind = CborVal::FLOAT; out.f = d; }
                else { out.kind = CborVal::SIMPLE; out.u = arg; }
            }
        }
        --depth;
        return ok;
    }
};

int parse_cbor(const char* path, CborVal& doc)
{
    std::vector<unsigned char> raw;
    if (read_file(path, raw)) return 1;
    CborIn in{raw.data(), raw.data() + raw.size()};
    if (!in.value(doc) || doc.kind != CborVal::MAP) return e2vq_set_error("%s: not a CBOR model document", path);
    return 0;
}

// ------------------------------------------------------------------------------------------
// sequences
// ------------------------------------------------------------------------------------------
struct Sequence {  // sequence::Sequence, src/sequence/mod.rs:10-15
    std::string class_name;
    uint32_t codebook_size = 0;
    std::vector<uint16_t> symbols;
};

int load_sequence(const char* path, Sequence& s)
{
    char cls[96];
    int M;
    int64_t T;
    if (e2vq_seq_info(path, cls, &M, &T)) return 1;
    s.class_name = cls;
    s.codebook_size = (uint32_t)M;
    s.symbols.resize((size_t)T);
    return T > 0 ? e2vq_seq_read(path, s.symbols.data(), T) : 0;
}

// ------------------------------------------------------------------------------------------
// nb -- src/nb/nbayes.rs
// ------------------------------------------------------------------------------------------
struct NBayes {  // :13-18
    std::string class_name;
    uint64_t total_symbols = 0;
    std::vector<uint64_t> frequencies;

    // probability of generating the symbol, using an m-estimate (:38-42)
    double prob_symbol(size_t symbol) const
    {
        const size_t codebook_size = frequencies.size();
        const double f = (double)frequencies[symbol];
        return (f + 1.0) / (double)(total_symbols + codebook_size);
    }
    double log_prob_symbol(size_t symbol) const { return log10(prob_symbol(symbol)); }  // :45-47
    double log_prob_sequence(const Sequence& seq) const  // :50-54: fold from 0.0 in symbol order
    {
        double acc = 0.0;
        for (uint16_t s : seq.symbols) acc = acc + log_prob_symbol(s);
        return acc;
    }
};

int nb_load(const char* path, NBayes& m)
{
    CborVal doc;
    if (parse_cbor(path, doc)) return 1;
    const CborVal *c = doc.get("class_name"), *t = doc.get("total_symbols"), *f = doc.get("frequencies");
    if (!c || c->kind != CborVal::TEXT || !t || t->kind != CborVal::UINT || !f || f->kind != CborVal::ARRAY)
        return e2vq_set_error("%s: not an NBayes model", path);
    m.class_name = c->s;
    m.total_symbols = t->u;
    m.frequencies.clear();
    for (const CborVal& v : f->items) {
        if (v.kind != CborVal::UINT) return e2vq_set_error("%s: not an NBayes model", path);
        m.frequencies.push_back(v.u);
    }
    return 0;
}

std::vector<unsigned char> nb_bytes(const NBayes& m)
{
    CborOut o;
    o.map(3);
    o.text("class_name");
    o.text(m.class_name);
    o.text("total_symbols");
    o.uint(m.total_symbols);
    o.text("frequencies");
    o.array(m.frequencies.size());
    for (uint64_t f : m.frequencies) o.uint(f);
    return o.b;
}

// ------------------------------------------------------------------------------------------
// mm -- src/mm/markov.rs
// ------------------------------------------------------------------------------------------
const float EQ_EPSILON = 1e-5f;  // :16

// ndarray's sum() of a contiguous f32 slice: numeric_util::unrolled_fold (eight partial sums), which is what the
// reference's row-stochastic asserts evaluate
float ndarray_sum(const float* xs, size_t n)
{
    float acc = 0.f, p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    while (n >= 8) {
        for (int k = 0; k < 8; ++k) p[k] = p[k] + xs[k];
        xs += 8;
        n -= 8;
    }
    acc = acc + (p[0] + p[4]);
    acc = acc + (p[1] + p[5]);
    acc = acc + (p[2] + p[6]);
    acc = acc + (p[3] + p[7]);
    for (size_t i = 0; i < n && i < 7; ++i) acc = acc + xs[i];
    return acc;
}

// assert_approx_eq!(a, b, eps): panics unless |a - b| < eps
bool approx_eq(float a, float b, float eps) { return fabsf(a - b) < eps; }

struct MM {  // :19-24
    std::string class_name;
    std::vector<float> pi;  // [M]
    std::vector<float> a;   // [M][M]
    size_t M() const { return pi.size(); }

    float log_prob_sequence(const Sequence& seq) const  // :43-49 (f32 throughout)
    {
        const size_t n = M();
        float p = log10f(pi[seq.symbols[0]]);
        for (size_t t = 0; t + 1 < seq.symbols.size(); ++t) p += log10f(a[(size_t)seq.symbols[t] * n + seq.symbols[t + 1]]);
        return p;
    }
};

bool read_ndarray(const CborVal* v, size_t ndim, std::vector<size_t>& dim, std::vector<float>& data)
{
    if (!v || v->kind != CborVal::MAP) return false;
    const CborVal *d = v->get("dim"), *x = v->get("data");
    if (!d || d->kind != CborVal::ARRAY || d->items.size() != ndim || !x || x->kind != CborVal::ARRAY) return false;
    size_t total = 1;
    dim.clear();
    for (const CborVal& k : d->items) {
        if (k.kind != CborVal::UINT) return false;
        dim.push_back((size_t)k.u);
        total *= (size_t)k.u;
    }
    if (x->items.size() != total) return false;
    data.clear();
    for (const CborVal& e : x->items) {
        double f;
        if (!e.number(&f)) return false;
        data.push_back((float)f);
    }
    return true;
}

int mm_load(const char* path, MM& m)
{
    CborVal doc;
    if (parse_cbor(path, doc)) return 1;
    const CborVal* c = doc.get("class_name");
    std::vector<size_t> d1, d2;
    if (!c || c->kind != CborVal::TEXT || !read_ndarray(doc.get("pi"), 1, d1, m.pi) ||
        !read_ndarray(doc.get("a"), 2, d2, m.a) || d2[0] != d1[0] || d2[1] != d1[0] || d1[0] < 1)
        return e2vq_set_error("%s: not an MM model", path);
    m.class_name = c->s;
    return 0;
}

void put_ndarray(CborOut& o, const std::vector<size_t>& dim, const std::vector<float>& data)
{
    o.map(3);  // ndarray's serde format, version 1
    o.text("v");
    o.uint(1);
    o.text("dim");
    o.array(dim.size());
    for (size_t d : dim) o.uint(d);
    o.text("data");
    o.array(data.size());
    for (float f : data) o.f32(f);
}

std::vector<unsigned char> mm_bytes(const MM& m)
{
    CborOut o;
    o.map(3);
    o.text("class_name");
    o.text(m.class_name);
    o.text("pi");
    put_ndarray(o, {m.M()}, m.pi);
    o.text("a");
    put_ndarray(o, {m.M(), m.M()}, m.a);
    return o.b;
}

// ndarray's Display of a 1-D f32 array: "[a, b, ...]", more than 11 elements abbreviated to 5 + "..." + 5
std::string ndarray_display_row(const float* x, size_t n)
{
    std::string s = "[";
    auto put = [&](size_t i) { s += rust_display(x[i]); };
    if (n <= 11) {
        for (size_t i = 0; i < n; ++i) { if (i) s += ", "; put(i); }
    } else {
        for (size_t i = 0; i < 5; ++i) { put(i); s += ", "; }
        s += "...";
        for (size_t i = n - 5; i < n; ++i) { s += ", "; put(i); }
    }
    return s + "]";
}

std::vector<std::string> to_strings(const char* const* v, int n)
{
    std::vector<std::string> out;
    for (int i = 0; i < n; ++i) out.emplace_back(v[i] ? v[i] : "");
    return out;
}

void copy_out(const std::string& s, char* out, int cap)
{
    if (out && cap > 0) snprintf(out, (size_t)cap, "%s", s.c_str());
}

}  // namespace

// ==========================================================================================
// .seq reader (sequence::load, src/sequence/mod.rs:49-75)
// ==========================================================================================
static int seq_open(const char* path, FILE** fp, char class_name[96], uint32_t* T, uint32_t* M)
{
    FILE* f = fopen(path, "rb");
    if (!f) return e2vq_set_error("%s: %s", path, strerror(errno));
    unsigned char hdr[120];
    if (fread(hdr, 1, sizeof hdr, f) != sizeof hdr) {
        fclose(f);
        return e2vq_set_error("%s: truncated header", path);
    }
    // read_file_ident + starts_with("<sequence>") (:53-56); class name up to the first NUL (utl/mod.rs:30-41)
    if (strncmp((const char*)hdr, "<sequence>", 10) != 0) {
        fclose(f);
        return e2vq_set_error("%s: Not a sequence", path);
    }
    memset(class_name, 0, 96);
    for (int i = 0; i < 95 && hdr[16 + i]; ++i) class_name[i] = (char)hdr[16 + i];
    *T = hdr[112] | (hdr[113] << 8) | (hdr[114] << 16) | ((uint32_t)hdr[115] << 24);
    *M = hdr[116] | (hdr[117] << 8) | (hdr[118] << 16) | ((uint32_t)hdr[119] << 24);
    *fp = f;
    return 0;
}

extern "C" int e2vq_seq_info(const char* path, char class_name[96], int* M, int64_t* T)
{
    FILE* f = nullptr;
    uint32_t t, m;
    if (seq_open(path, &f, class_name, &t, &m)) return 1;
    fclose(f);
    *M = (int)m;
    *T = (int64_t)t;
    return 0;
}

extern "C" int e2vq_seq_read(const char* path, uint16_t* sym, int64_t capacity)
{
    FILE* f = nullptr;
    char cls[96];
    uint32_t t, m;
    if (seq_open(path, &f, cls, &t, &m)) return 1;
    int rc = 0;
    if ((int64_t)t > capacity) rc = e2vq_set_error("%s: %u symbols exceed the buffer", path, t);
    if (!rc) {
        std::vector<unsigned char> raw((size_t)t * 2);
        if (fread(raw.data(), 1, raw.size(), f) != raw.size()) rc = e2vq_set_error("%s: truncated payload", path);
        for (uint32_t i = 0; !rc && i < t; ++i) sym[i] = (uint16_t)(raw[2 * (size_t)i] | (raw[2 * (size_t)i + 1] << 8));
    }
    fclose(f);
    return rc;
}

// ==========================================================================================
// c12n driven directly (tests, hmm classify)
// ==========================================================================================
extern "C" int e2vq_c12n_run(const char* const* model_class_names, int num_models, const int* class_ids,
                             const char* const* case_class_names, const char* const* case_titles, const double* probs,
                             int num_cases, int show_ranked, const char* out_base_name, int* result, int* confusion)
{
    FlushStdout flush_on_return;
    if (num_models < 1 || num_cases < 0) return e2vq_set_error("e2vq_c12n_run: bad arguments");
    C12nResults c(to_strings(model_class_names, num_models));
    for (int k = 0; k < num_cases; ++k) {
        if (class_ids[k] < 0 || class_ids[k] >= num_models) return e2vq_set_error("case %d: class id out of range", k);
        std::vector<double> p(probs + (size_t)k * num_models, probs + (size_t)(k + 1) * num_models);
        c.add_case((size_t)class_ids[k], case_class_names[k], p, show_ranked != 0,
                   [&] { return std::string(case_titles ? case_titles[k] : ""); });
    }
    printf("\n");
    int rc = c.report_results(c.model_class_names, out_base_name);
    const int n1 = num_models + 1;
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n1; ++j) {
            if (result) result[i * n1 + j] = c.result[i][j];
            if (confusion) confusion[i * n1 + j] = c.confusion[i][j];
        }
    return rc;
}

// ==========================================================================================
// nb
// ==========================================================================================
extern "C" int ecoz2_nb_learn(int codebook_size, const char* const* seq_filenames, int num_sequences, char* out_path,
                              int out_path_cap)
{
    FlushStdout flush_on_return;
    if (!seq_filenames || num_sequences < 1 || codebook_size < 1) return e2vq_set_error("ecoz2_nb_learn: bad arguments");
    // nbayes::learn, src/nb/nbayes.rs:63-114
    Sequence seq;
    if (load_sequence(seq_filenames[0], seq)) return 1;  // class name (and model size) from the first sequence
    NBayes model;
    model.class_name = seq.class_name;
    printf("NB learn: num sequences=%d class='%s' codebook_size=%d\n", num_sequences, model.class_name.c_str(),
           codebook_size);
    model.frequencies.assign(seq.codebook_size, 0);
    for (int i = 0; i < num_sequences; ++i) {
        if (load_sequence(seq_filenames[i], seq)) return 1;
        fputs(coloured(".", 35).c_str(), stdout);
        fflush(stdout);
        if ((uint32_t)codebook_size != seq.codebook_size)
            return e2vq_set_error("conformity error: codebook size: %d != %u", codebook_size, seq.codebook_size);
        if (model.class_name != seq.class_name)
            return e2vq_set_error("conformity error: class_name: %s != %s", model.class_name.c_str(), seq.class_name.c_str());
        model.total_symbols += seq.symbols.size();
        for (uint16_t s : seq.symbols) {
            if (s >= model.frequencies.size()) return e2vq_set_error("%s: symbol %u out of range", seq_filenames[i], s);
            model.frequencies[s] += 1;
        }
    }
    printf("\n");
    // main_nbayes_learn, src/nb/mod.rs:116-126
    char filename[4096];
    snprintf(filename, sizeof filename, "data/nbs/M%d/%s.nb", codebook_size, model.class_name.c_str());
    printf("NB model trained\n");
    if (write_file(std::string(out_root()) + "/" + filename, nb_bytes(model))) return 1;
    printf("NB model saved: %s\n\n\n", filename);
    copy_out(std::string(out_root()) + "/" + filename, out_path, out_path_cap);
    return 0;
}

extern "C" int e2vq_nb_log_prob(const char* nb_filename, const char* seq_filename, double* log_prob)
{
    NBayes m;
    Sequence s;
    if (nb_load(nb_filename, m) || load_sequence(seq_filename, s)) return 1;
    for (uint16_t v : s.symbols)
        if (v >= m.frequencies.size()) return e2vq_set_error("%s: symbol %u out of the model's range", seq_filename, v);
    *log_prob = m.log_prob_sequence(s);
    return 0;
}

extern "C" int ecoz2_nb_classify(const char* const* nb_filenames, int num_models, const char* const* seq_filenames,
                                 int num_sequences, int show_ranked, int codebook_size)
{
    FlushStdout flush_on_return;
    if (!nb_filenames || num_models < 1 || !seq_filenames || num_sequences < 0)
        return e2vq_set_error("ecoz2_nb_classify: bad arguments");
    // nbayes::classify, src/nb/nbayes.rs:116-153
    printf("Loading NBayes models\n");
    std::vector<NBayes> models((size_t)num_models);
    std::vector<std::string> names;
    for (int i = 0; i < num_models; ++i) {
        if (nb_load(nb_filenames[i], models[i])) return 1;
        names.push_back(models[i].class_name);
    }
    C12nResults c12n(names);
    printf("Classifying sequences\n");
    for (int k = 0; k < num_sequences; ++k) {
        Sequence seq;
        if (load_sequence(seq_filenames[k], seq)) return 1;
        const auto it = std::find(names.begin(), names.end(), seq.class_name);
        if (it == names.end()) continue;  // no model of that class: the sequence is skipped (:138)
        std::vector<double> probs;
        for (const NBayes& m : models) {
            for (uint16_t v : seq.symbols)
                if (v >= m.frequencies.size()) return e2vq_set_error("%s: symbol %u out of the model's range", seq_filenames[k], v);
            probs.push_back(m.log_prob_sequence(seq));
        }
        c12n.add_case((size_t)(it - names.begin()), seq.class_name, probs, show_ranked != 0, [&] {
            return std::string("\n") + seq_filenames[k] + ": '" + seq.class_name + "'\n";
        });
    }
    printf("\n");
    return c12n.report_results(names, "nb_" + std::to_string(codebook_size));
}

extern "C" int ecoz2_nb_show(const char* nb_filename)
{
    FlushStdout flush_on_return;
    NBayes m;
    if (nb_load(nb_filename, m)) return 1;
    // NBayes::show, src/nb/nbayes.rs:21-35
    printf("# class_name='%s', M=%zu total_symbols=%llu\n", m.class_name.c_str(), m.frequencies.size(),
           (unsigned long long)m.total_symbols);
    printf("%-4s, %-4s, prob\n", "m", "frequency");
    for (size_t s = 0; s < m.frequencies.size(); ++s)
        printf("%4zu, %4llu, %.7f\n", s, (unsigned long long)m.frequencies[s], m.prob_symbol(s));
    return 0;
}

// ==========================================================================================
// mm
// ==========================================================================================
extern "C" int ecoz2_mm_learn(int codebook_size, const char* const* seq_filenames, int num_sequences, char* out_path,
                              int out_path_cap)
{
    FlushStdout flush_on_return;
    if (!seq_filenames || num_sequences < 1 || codebook_size < 1) return e2vq_set_error("ecoz2_mm_learn: bad arguments");
    // markov::learn, src/mm/markov.rs:59-126
    Sequence seq;
    if (load_sequence(seq_filenames[0], seq)) return 1;
    MM model;
    model.class_name = seq.class_name;
    printf("MM learn: num sequences=%d class='%s' codebook_size=%d\n", num_sequences, model.class_name.c_str(),
           codebook_size);
    const size_t n = (size_t)codebook_size;
    // init counters: pi and a are initially just counters (:70-74)
    model.pi.assign(n, 1.f);
    std::vector<int> n_js(n, 0);
    model.a.assign(n * n, 1.f);
    for (int i = 0; i < num_sequences; ++i) {
        if (load_sequence(seq_filenames[i], seq)) return 1;
        fputs(coloured(".", 35).c_str(), stdout);
        fflush(stdout);
        if ((uint32_t)codebook_size != seq.codebook_size)
            return e2vq_set_error("conformity error: codebook size: %d != %u", codebook_size, seq.codebook_size);
        if (model.class_name != seq.class_name)
            return e2vq_set_error("conformity error: class_name: %s != %s", model.class_name.c_str(), seq.class_name.c_str());
        if (seq.symbols.empty()) return e2vq_set_error("%s: empty sequence", seq_filenames[i]);  // (seq.symbols[0] panics)
        for (uint16_t s : seq.symbols)
            if (s >= n) return e2vq_set_error("%s: symbol %u out of range", seq_filenames[i], s);
        // update counts (:102-108)
        model.pi[seq.symbols[0]] += 1.f;
        for (size_t t = 0; t + 1 < seq.symbols.size(); ++t) {
            const size_t j = seq.symbols[t], k = seq.symbols[t + 1];
            n_js[j] += 1;                  // one more transition from symbol j
            model.a[j * n + k] += 1.f;     // one more j->k transition
        }
    }
    printf("\n");
    const float num_seqs = (float)num_sequences;
    // normalize pi (:115-117)
    {
        const float d = num_seqs + (float)codebook_size;
        for (float& v : model.pi) v = v / d;
        if (!approx_eq(ndarray_sum(model.pi.data(), n), 1.f, EQ_EPSILON))
            return e2vq_set_error("assertion failed: pi.sum() = %g is not 1 within %g", (double)ndarray_sum(model.pi.data(), n),
                                  (double)EQ_EPSILON);
    }
    // normalize rows in a (:119-123)
    for (size_t j = 0; j < n; ++j) {
        const float d = (float)n_js[j] + (float)codebook_size;
        float* row = &model.a[j * n];
        for (size_t k = 0; k < n; ++k) row[k] = row[k] / d;
        if (!approx_eq(ndarray_sum(row, n), 1.f, EQ_EPSILON))
            return e2vq_set_error("assertion failed: row %zu of A sums to %g, not 1 within %g", j, (double)ndarray_sum(row, n),
                                  (double)EQ_EPSILON);
    }
    // main_mm_learn, src/mm/mod.rs:116-126
    char filename[4096];
    snprintf(filename, sizeof filename, "data/mms/M%d/%s.mm", codebook_size, model.class_name.c_str());
    printf("MM model trained\n");
    if (write_file(std::string(out_root()) + "/" + filename, mm_bytes(model))) return 1;
    printf("MM model saved: %s\n\n\n", filename);
    copy_out(std::string(out_root()) + "/" + filename, out_path, out_path_cap);
    return 0;
}

static int mm_check_symbols(const MM& m, const Sequence& s, const char* file)
{
    if (s.symbols.empty()) return e2vq_set_error("%s: empty sequence", file);
    for (uint16_t v : s.symbols)
        if (v >= m.M()) return e2vq_set_error("%s: symbol %u out of the model's range", file, v);
    return 0;
}

extern "C" int e2vq_mm_log_prob(const char* mm_filename, const char* seq_filename, float* log_prob)
{
    MM m;
    Sequence s;
    if (mm_load(mm_filename, m) || load_sequence(seq_filename, s) || mm_check_symbols(m, s, seq_filename)) return 1;
    *log_prob = m.log_prob_sequence(s);
    return 0;
}

extern "C" int ecoz2_mm_classify(const char* const* mm_filenames, int num_models, const char* const* seq_filenames,
                                 int num_sequences, int show_ranked, int codebook_size)
{
    FlushStdout flush_on_return;
    if (!mm_filenames || num_models < 1 || !seq_filenames || num_sequences < 0)
        return e2vq_set_error("ecoz2_mm_classify: bad arguments");
    // markov::classify, src/mm/markov.rs:128-167
    printf("Loading MM models\n");
    std::vector<MM> models((size_t)num_models);
    std::vector<std::string> names;
    for (int i = 0; i < num_models; ++i) {
        if (mm_load(mm_filenames[i], models[i])) return 1;
        names.push_back(models[i].class_name);
    }
    C12nResults c12n(names);
    printf("Classifying sequences\n");
    for (int k = 0; k < num_sequences; ++k) {
        Sequence seq;
        if (load_sequence(seq_filenames[k], seq)) return 1;
        const auto it = std::find(names.begin(), names.end(), seq.class_name);
        if (it == names.end()) continue;
        std::vector<double> probs;
        for (const MM& m : models) {
            if (mm_check_symbols(m, seq, seq_filenames[k])) return 1;
            probs.push_back((double)m.log_prob_sequence(seq));  // `as f64` (:150)
        }
        c12n.add_case((size_t)(it - names.begin()), seq.class_name, probs, show_ranked != 0,
                      [&] { return std::string("\n") + seq_filenames[k] + ": '" + seq.class_name + "'"; });
    }
    printf("\n");
    return c12n.report_results(names, "mm_" + std::to_string(codebook_size));
}

extern "C" int ecoz2_mm_show(const char* mm_filename)
{
    FlushStdout flush_on_return;
    MM m;
    if (mm_load(mm_filename, m)) return 1;
    // MM::show, src/mm/markov.rs:27-40
    const size_t n = m.M();
    printf("class_name='%s', codebook_size=%zu\n", m.class_name.c_str(), n);
    printf("pi = %s\n", ndarray_display_row(m.pi.data(), n).c_str());
    if (!approx_eq(ndarray_sum(m.pi.data(), n), 1.f, EQ_EPSILON)) return e2vq_set_error("assertion failed: pi does not sum to 1");
    printf("A =\n");
    for (size_t j = 0; j < n; ++j) {
        printf(" [%zu]: %s\n", j, ndarray_display_row(&m.a[j * n], n).c_str());
        if (!approx_eq(ndarray_sum(&m.a[j * n], n), 1.f, EQ_EPSILON))
            return e2vq_set_error("assertion failed: row %zu of A does not sum to 1", j);
    }
    return 0;
}
