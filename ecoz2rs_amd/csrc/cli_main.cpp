// cli_main.cpp -- `ecoz2 vq {learn,quantize,show}` front-end over libecoz2vq.so.
// Mirrors the reference's clap option structs for this path
// (/root/reference/src/vq/mod.rs:38-96 VqLearnOpts / VqQuantizeOpts, :120-133 VqShowOpts)
// and its mains (:151-216): same flag names, defaults, file-list resolution and messages.
// Like the reference (src/vq/mod.rs:146-148) errors are printed and the exit code stays 0
// unless the arguments themselves are unusable.
#include "../../include/ecoz2_classify.h"
#include "../../include/ecoz2_vq.h"
#include "vq_io.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

static void callback(void*, int M, double avg, double sigma, double inertia)
{
    // Ecoz2ObserverRef::step, src/ecoz2_lib/mod.rs:61-69
    printf("   Ecoz2ObserverRef.step: M=%d avg_distortion=%g sigma=%g inertia=%g\n", M, avg, sigma, inertia);
}

static int usage()
{
    fprintf(stderr,
            "usage:\n"
            "  ecoz2 vq learn [-B <codebook>] [-P <P>] [-e <eps>] [--class-name <class>] [--exp-key <k>]\n"
            "                 --predictors <files|dirs|tt.csv>...\n"
            "  ecoz2 vq quantize --codebook <cbook> --predictors <files|dirs|tt.csv>...\n"
            "                 [--predictors-dir-template <t>] [--tt <TRAIN|TEST>] [--class-name <class>] [-s]\n"
            "  ecoz2 vq classify [-r] --codebooks <files|dirs>... --tt <TRAIN|TEST> --predictors <files|dirs|tt.csv>...\n"
            "  ecoz2 vq show [-f <from>] [-t <to>] <codebook>\n"
            "  ecoz2 seq show [-c] [-L] [--full] [--pickle out.pkl -M <M> --tt <TRAIN|TEST> [--class-name c]] <file.seq|tt.csv>...\n"
            "  ecoz2 prd show [-k] [--from a] [--to b] <file.prd>\n"
            "  ecoz2 {nb|mm} learn -M <M> [--class-name <class>] <file.seq|dirs|tt.csv>...\n"
            "  ecoz2 {nb|mm} classify -M <M> [-r] --tt <TRAIN|TEST> --models <files|dirs>... --sequences <files|dirs|tt.csv>...\n"
            "  ecoz2 {nb|mm} show --model <file>\n"
            "  ecoz2 hmm learn [-N 5] -M <M> [-t 3] [-I -1] [-e 1e-05] [-a 0.3] [-s <seed>] [--ser] [--class-name c]\n"
            "                  --sequences <file.seq|dirs|tt.csv>...\n"
            "  ecoz2 hmm classify [-r] [-c|--c12n <out.csv>] -m|--models <files|dirs>... --tt <TRAIN|TEST> -M <M> [--class-name c]\n"
            "                  (-s|--sequences <files|dirs|tt.csv>... | --predictors <files|dirs|tt.csv>... --codebooks <files|dirs>...\n"
            "                   [--predictors-dir-template <t>])\n"
            "  ecoz2 hmm show --hmm <file> [-f|--format \"%%Lg \"]\n"
            "  ecoz2 cversion\n");
    return 2;
}

static bool is_flag(const char* a) { return a[0] == '-' && a[1] != 0 && !(a[1] >= '0' && a[1] <= '9'); }

static std::vector<const char*> cptrs(const std::vector<std::string>& v)
{
    std::vector<const char*> p;
    for (const auto& s : v) p.push_back(s.c_str());
    return p;
}

static int vq_learn(int argc, char** argv)
{
    std::string base, cls, exp_key;
    int P = -1;
    double eps = 0.05;
    std::vector<std::string> predictors;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", name); exit(2); }
            return argv[++i];
        };
        if (a == "-B" || a == "--base-codebook") base = val("-B");
        else if (a == "-P" || a == "--prediction-order") P = atoi(val("-P"));
        else if (a == "-e" || a == "--epsilon") eps = atof(val("-e"));
        else if (a == "--class-name") cls = val("--class-name");
        else if (a == "--exp-key") exp_key = val("--exp-key");
        else if (a == "--predictors") { while (i + 1 < argc && !is_flag(argv[i + 1])) predictors.push_back(argv[++i]); }
        else if (!is_flag(argv[i])) predictors.push_back(a);
        else return usage();
    }
    if (!base.empty() && P >= 0) {  // src/vq/mod.rs:161-163
        printf("Only one of base codebook or prediction order expected\n");
        return 0;
    }
    if (base.empty() && P < 0) return usage();
    const std::string codebook_class = cls.empty() ? "_" : cls;
    std::vector<std::string> files;
    const bool tt_list = predictors.size() == 1 && predictors[0].size() > 4 &&
                         predictors[0].compare(predictors[0].size() - 4, 4, ".csv") == 0;
    int rc = tt_list ? e2vq_io::files_from_csv(predictors[0], "TRAIN", cls, "predictors", ".prd", nullptr, files)
                     : e2vq_io::resolve_filenames(predictors, ".prd", files);
    if (!rc && files.empty()) { printf("No predictors given\n"); return 0; }
    if (rc) { printf("%s\n", e2vq_last_error()); return 0; }
    printf("vq_learn: base_codebook_opt=%s prediction_order=%d, epsilon=%g codebook_class_name=%s predictor_filenames: %zu\n",
           base.empty() ? "None" : base.c_str(), P, eps, codebook_class.c_str(), files.size());
    auto ptrs = cptrs(files);
    if (!base.empty())
        ecoz2_vq_learn_using_base_codebook(base.c_str(), eps, ptrs.data(), (int)ptrs.size(), nullptr, callback);
    else
        ecoz2_vq_learn(P, eps, codebook_class.c_str(), ptrs.data(), (int)ptrs.size(), nullptr, callback);
    return 0;
}

static int vq_quantize(int argc, char** argv)
{
    std::string codebook, tmpl = "data/predictors", tt, cls;
    bool show = false;
    std::vector<std::string> predictors;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", name); exit(2); }
            return argv[++i];
        };
        if (a == "--codebook") codebook = val("--codebook");
        else if (a == "--predictors-dir-template") tmpl = val("--predictors-dir-template");
        else if (a == "--tt") tt = val("--tt");
        else if (a == "--class-name") cls = val("--class-name");
        else if (a == "-s" || a == "--show-filenames") show = true;
        else if (a == "--predictors") { while (i + 1 < argc && !is_flag(argv[i + 1])) predictors.push_back(argv[++i]); }
        else if (!is_flag(argv[i])) predictors.push_back(a);
        else return usage();
    }
    if (codebook.empty() || predictors.empty()) return usage();
    std::vector<std::string> files;
    const bool tt_list = predictors.size() == 1 && predictors[0].size() > 4 &&
                         predictors[0].compare(predictors[0].size() - 4, 4, ".csv") == 0;
    int rc = tt_list ? e2vq_io::files_from_csv(predictors[0], tt, cls, "", ".prd", &tmpl, files)
                     : e2vq_io::resolve_filenames(predictors, ".prd", files);
    if (rc) { printf("%s\n", e2vq_last_error()); return 0; }
    printf("number of predictor files: %zu\n", files.size());  // src/vq/mod.rs:211
    printf("nom_raas = %s\n", codebook.c_str());               // src/ecoz2_lib/mod.rs:326
    auto ptrs = cptrs(files);
    ecoz2_vq_quantize(codebook.c_str(), ptrs.data(), (int)ptrs.size(), show ? 1 : 0);
    return 0;
}

static int vq_classify(int argc, char** argv)
{
    bool ranked = false;
    std::string tt;
    std::vector<std::string> codebooks, predictors;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-r" || a == "--show-ranked") ranked = true;
        else if (a == "--tt" && i + 1 < argc) tt = argv[++i];
        else if (a == "--codebooks") { while (i + 1 < argc && !is_flag(argv[i + 1])) codebooks.push_back(argv[++i]); }
        else if (a == "--predictors") { while (i + 1 < argc && !is_flag(argv[i + 1])) predictors.push_back(argv[++i]); }
        else return usage();
    }
    if (codebooks.empty() || predictors.empty() || tt.empty()) return usage();
    std::vector<std::string> cbs, prds;
    e2vq_io::resolve_filenames(codebooks, ".cbook", cbs);
    if (cbs.empty()) { printf("No codebooks given\n"); return 0; }
    const bool tt_list = predictors.size() == 1 && predictors[0].size() > 4 &&
                         predictors[0].compare(predictors[0].size() - 4, 4, ".csv") == 0;
    int rc = tt_list ? e2vq_io::files_from_csv(predictors[0], tt, "", "predictors", ".prd", nullptr, prds)
                     : e2vq_io::resolve_filenames(predictors, ".prd", prds);
    if (rc) { printf("%s\n", e2vq_last_error()); return 0; }
    if (prds.empty()) { printf("No predictors given\n"); return 0; }
    printf("number of codebooks: %zu  number of predictors: %zu\n", cbs.size(), prds.size());  // src/vq/mod.rs:236-241
    printf("show_ranked = %s\n", ranked ? "true" : "false");
    auto pc = cptrs(cbs), pp = cptrs(prds);
    ecoz2_vq_classify(pc.data(), (int)pc.size(), pp.data(), (int)pp.size(), ranked ? 1 : 0);
    return 0;
}

static int vq_show(int argc, char** argv)
{
    int from = -1, to = -1;
    std::string codebook;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if ((a == "-f" || a == "--from") && i + 1 < argc) from = atoi(argv[++i]);
        else if ((a == "-t" || a == "--to") && i + 1 < argc) to = atoi(argv[++i]);
        else if (a == "--codebook" && i + 1 < argc) codebook = argv[++i];
        else codebook = a;
    }
    if (codebook.empty()) return usage();
    printf("codebook_filename = %s\n", codebook.c_str());  // src/ecoz2_lib/mod.rs:361
    ecoz2_vq_show(codebook.c_str(), from, to);
    return 0;
}

// `ecoz2 seq show [-c] [-L] [--full] <files...>`: Sequence::show, /root/reference/src/sequence/mod.rs:17-47
// (reads the C-format .seq exactly as Sequence::load does, :49-75)
static bool load_seq(const std::string& f, std::string& cls, unsigned& M, std::vector<unsigned>& sym)
{
    FILE* fp = fopen(f.c_str(), "rb");
    unsigned char hdr[120];
    if (!fp || fread(hdr, 1, sizeof hdr, fp) != sizeof hdr || strncmp((const char*)hdr, "<sequence>", 10) != 0) {
        if (fp) fclose(fp);
        return false;
    }
    char c[97] = {0};
    memcpy(c, hdr + 16, 96);
    cls = c;
    const unsigned len = hdr[112] | (hdr[113] << 8) | (hdr[114] << 16) | ((unsigned)hdr[115] << 24);
    M = hdr[116] | (hdr[117] << 8) | (hdr[118] << 16) | ((unsigned)hdr[119] << 24);
    sym.assign(len, 0);
    for (unsigned t = 0; t < len; ++t) {
        unsigned char b[2];
        if (fread(b, 1, 2, fp) != 2) break;
        sym[t] = b[0] | (b[1] << 8);
    }
    fclose(fp);
    return true;
}

// pickle (protocol 2) of a list of lists of ints: what `utl::to_pickle(&list_of_sequences, ..)` exports
// (/root/reference/src/seq/mod.rs:88-112, src/utl/mod.rs:277-283); loads with Python's pickle.load
static bool write_pickle(const std::string& path, const std::vector<std::vector<unsigned>>& seqs)
{
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp) return false;
    auto put_int = [&](unsigned v) {
        if (v < 256) { fputc('K', fp); fputc((int)v, fp); }                                   // BININT1
        else if (v < 65536) { fputc('M', fp); fputc(v & 255, fp); fputc(v >> 8, fp); }        // BININT2
        else { fputc('J', fp); for (int k = 0; k < 4; ++k) fputc((v >> (8 * k)) & 255, fp); } // BININT
    };
    fputc(0x80, fp); fputc(2, fp);  // PROTO 2
    fputc(']', fp);                 // EMPTY_LIST
    fputc('(', fp);                 // MARK
    for (const auto& s : seqs) {
        fputc(']', fp);
        fputc('(', fp);
        for (unsigned v : s) put_int(v);
        fputc('e', fp);             // APPENDS
    }
    fputc('e', fp);
    fputc('.', fp);                 // STOP
    return fclose(fp) == 0;
}

static int seq_show(int argc, char** argv)
{
    bool no_sequence = false, only_length = false, full = false;
    std::string pickle, cls_filter, tt;
    int codebook_size = -1;
    std::vector<std::string> files;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-c") no_sequence = true;
        else if (a == "-L") only_length = true;
        else if (a == "--full") full = true;
        else if (a == "--pickle" && i + 1 < argc) pickle = argv[++i];
        else if (a == "--class-name" && i + 1 < argc) cls_filter = argv[++i];
        else if (a == "--tt" && i + 1 < argc) tt = argv[++i];
        else if ((a == "-M" || a == "--codebook-size") && i + 1 < argc) codebook_size = atoi(argv[++i]);
        else if (!is_flag(argv[i])) files.push_back(a);
        else return usage();
    }
    if (files.empty()) return usage();
    if (!pickle.empty()) {  // src/seq/mod.rs:88-118
        if (codebook_size < 0 || tt.empty()) {
            printf("--codebook-size and --tt required when --pickle given\n");
            return 0;
        }
        std::vector<std::string> seq_files;
        const bool tt_list = files.size() == 1 && files[0].size() > 4 && files[0].compare(files[0].size() - 4, 4, ".csv") == 0;
        const std::string subdir = "sequences/M" + std::to_string(codebook_size);
        int rc = tt_list ? e2vq_io::files_from_csv(files[0], tt, cls_filter, subdir, ".seq", nullptr, seq_files)
                         : e2vq_io::resolve_filenames(files, ".seq", seq_files);
        if (rc) { printf("%s\n", e2vq_last_error()); return 0; }
        std::vector<std::vector<unsigned>> seqs;
        for (const auto& f : seq_files) {
            std::string cls; unsigned M; std::vector<unsigned> sym;
            if (!load_seq(f, cls, M, sym)) { printf("%s: Not a sequence\n", f.c_str()); return 0; }
            seqs.push_back(sym);
        }
        if (!write_pickle(pickle, seqs)) { printf("%s: cannot write\n", pickle.c_str()); return 0; }
        printf("%zu sequence(s) saved to \"%s\"\n", seqs.size(), pickle.c_str());
        return 0;
    }
    for (const auto& f : files) {
        std::string cls_s;
        unsigned M = 0;
        std::vector<unsigned> sym;
        if (!load_seq(f, cls_s, M, sym)) {
            printf("%s: Not a sequence\n", f.c_str());
            continue;
        }
        const char* cls = cls_s.c_str();
        const unsigned len = (unsigned)sym.size();
        if (no_sequence) continue;
        if (only_length) { printf("%u\n", len); continue; }
        printf("<%s(M=%u,L=%u): ", cls, M, len);
        if (full || len <= 30) {
            for (unsigned t = 0; t < len; ++t) printf("%s%u", t ? ", " : "", sym[t]);
        } else {
            for (unsigned t = 0; t < 10; ++t) printf("%s%u", t ? ", " : "", sym[t]);
            printf(", ..., ");
            for (unsigned t = len - 10; t < len; ++t) printf("%s%u", t > len - 10 ? ", " : "", sym[t]);
        }
        printf(">\n");
    }
    return 0;
}

// `ecoz2 prd show [-k|--reflections] [-f|--from a] [-t|--to b] <file>` (options: src/prd/mod.rs:30-62; from defaults to 1,
// to = 0 means P) -> ecoz2_prd_show_file, the symbol the reference binds (src/ecoz2_lib/mod.rs:89-94, src/prd/mod.rs:99)
static int prd_show(int argc, char** argv)
{
    int from = 1, to = 0, refl = 0;
    std::string file;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if ((a == "--from" || a == "-f") && i + 1 < argc) from = atoi(argv[++i]);
        else if ((a == "--to" || a == "-t") && i + 1 < argc) to = atoi(argv[++i]);
        else if (a == "-k" || a == "--reflections") refl = 1;
        else if (!is_flag(argv[i])) file = a;
        else return usage();
    }
    if (file.empty()) return usage();
    ecoz2_prd_show_file(file.c_str(), refl, from, to);
    return 0;
}

// `ecoz2 nb ...` / `ecoz2 mm ...`: option structs and mains of /root/reference/src/nb/mod.rs:32-161 and
// src/mm/mod.rs:32-161 (identical shapes; the models differ)
static bool is_csv_list(const std::vector<std::string>& v)
{
    return v.size() == 1 && v[0].size() > 4 && v[0].compare(v[0].size() - 4, 4, ".csv") == 0;
}

static int seq_model_cmd(bool nb, int argc, char** argv)
{
    if (argc < 1) return usage();
    const std::string cmd = argv[0];
    const char* ext = nb ? ".nb" : ".mm";
    int M = -1;
    bool ranked = false;
    std::string cls, tt, model;
    std::vector<std::string> models, sequences;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if ((a == "-M" || a == "--codebook-size") && i + 1 < argc) M = atoi(argv[++i]);
        else if ((a == "--class-name") && i + 1 < argc) cls = argv[++i];
        else if (a == "-r" || a == "--show-ranked") ranked = true;
        else if (a == "--tt" && i + 1 < argc) tt = argv[++i];
        else if ((a == "-m" || a == "--model") && i + 1 < argc) model = argv[++i];
        else if (a == "--models") { while (i + 1 < argc && !is_flag(argv[i + 1])) models.push_back(argv[++i]); }
        else if (a == "--sequences") { while (i + 1 < argc && !is_flag(argv[i + 1])) sequences.push_back(argv[++i]); }
        else if (!is_flag(argv[i])) sequences.push_back(a);
        else return usage();
    }
    if (cmd == "show") {
        if (model.empty()) return usage();
        if (nb ? ecoz2_nb_show(model.c_str()) : ecoz2_mm_show(model.c_str())) printf("%s\n", e2vq_last_error());
        return 0;
    }
    if (M < 1 || sequences.empty()) return usage();
    const std::string subdir = "sequences/M" + std::to_string(M);
    std::vector<std::string> seq_files;
    if (cmd == "learn") {  // main_nbayes_learn / main_mm_learn: resolve_files(sequences, "TRAIN", class_name, ..)
        int rc = is_csv_list(sequences) ? e2vq_io::files_from_csv(sequences[0], "TRAIN", cls, subdir, ".seq", nullptr, seq_files)
                                        : e2vq_io::resolve_filenames(sequences, ".seq", seq_files);
        if (rc || seq_files.empty()) { printf("%s\n", rc ? e2vq_last_error() : "No sequences given"); return 0; }
        auto ps = cptrs(seq_files);
        if (nb ? ecoz2_nb_learn(M, ps.data(), (int)ps.size(), nullptr, 0) : ecoz2_mm_learn(M, ps.data(), (int)ps.size(), nullptr, 0))
            printf("%s\n", e2vq_last_error());
        return 0;
    }
    if (cmd == "classify") {  // main_nbayes_classify / main_mm_classify
        if (tt.empty() || models.empty()) return usage();
        std::vector<std::string> model_files;
        e2vq_io::resolve_filenames(models, ext, model_files);
        if (model_files.empty()) { printf("No models given\n"); return 0; }
        int rc = is_csv_list(sequences) ? e2vq_io::files_from_csv(sequences[0], tt, "", subdir, ".seq", nullptr, seq_files)
                                        : e2vq_io::resolve_filenames(sequences, ".seq", seq_files);
        if (rc) { printf("%s\n", e2vq_last_error()); return 0; }
        printf("number of %s models: %zu  number of sequences: %zu\n", nb ? "NBayes" : "MM", model_files.size(), seq_files.size());
        printf("show_ranked = %s\n", ranked ? "true" : "false");
        auto pm = cptrs(model_files), ps = cptrs(seq_files);
        if (nb ? ecoz2_nb_classify(pm.data(), (int)pm.size(), ps.data(), (int)ps.size(), ranked, M)
               : ecoz2_mm_classify(pm.data(), (int)pm.size(), ps.data(), (int)ps.size(), ranked, M))
            printf("%s\n", e2vq_last_error());
        return 0;
    }
    return usage();
}

// `ecoz2 hmm {learn,classify,show}`: option structs and mains of /root/reference/src/hmm/mod.rs:40-291
static void hmm_callback(char*, double) {}  // the reference's Rust callback is a no-op too (src/hmm/mod.rs:205-207)

static int hmm_cmd(int argc, char** argv)
{
    if (argc < 1) return usage();
    const std::string cmd = argv[0];
    int N = 5, M = -1, type = 3, max_iterations = -1;
    double epsilon = 1e-05, val_auto = 0.3;
    long seed = -1;
    bool ser = false, ranked = false;
    std::string cls, tt, c12n, hmm, format = "%Lg ", tmpl = "data/predictors";
    std::vector<std::string> sequences, models, predictors, codebooks;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", name); exit(2); }
            return argv[++i];
        };
        auto many = [&](std::vector<std::string>& v) { while (i + 1 < argc && !is_flag(argv[i + 1])) v.push_back(argv[++i]); };
        if (a == "-N" || a == "--num-states") N = atoi(val("-N"));
        else if (a == "-M" || a == "--codebook-size") M = atoi(val("-M"));
        else if (a == "-t") type = atoi(val("-t"));
        else if (a == "-I" || a == "--max-iterations") max_iterations = atoi(val("-I"));
        else if (a == "-e") epsilon = atof(val("-e"));
        else if (a == "-a") val_auto = atof(val("-a"));
        else if (cmd == "learn" && (a == "-s" || a == "--seed")) seed = atol(val("-s"));
        else if (a == "--ser") ser = true;
        else if (a == "--class-name") cls = val("--class-name");
        else if (a == "-r" || a == "--show-ranked") ranked = true;
        else if (a == "-c" || a == "--c12n") c12n = val("--c12n");
        else if (a == "--tt") tt = val("--tt");
        else if (a == "-m" || a == "--models") many(models);
        else if (a == "-s" || a == "--sequences") many(sequences);
        else if (a == "--predictors") many(predictors);
        else if (a == "--predictors-dir-template") tmpl = val("--predictors-dir-template");
        else if (a == "--codebooks") many(codebooks);
        else if (a == "--hmm") hmm = val("--hmm");
        else if (a == "-f" || a == "--format") format = val("--format");
        else if (!is_flag(argv[i])) sequences.push_back(a);
        else return usage();
    }
    if (cmd == "show") {  // main_hmm_show
        if (hmm.empty()) return usage();
        printf("hmm_show: hmm_filename=%s format=%s\n", hmm.c_str(), format.c_str());  // src/ecoz2_lib/mod.rs:482-486
        if (ecoz2_hmm_show(hmm.c_str(), format.c_str())) printf("%s\n", e2vq_last_error());
        return 0;
    }
    if (M < 1) return usage();
    const std::string subdir = "sequences/M" + std::to_string(M);
    if (cmd == "learn") {  // main_hmm_learn, src/hmm/mod.rs:172-217
        std::vector<std::string> seq_files;
        int rc = is_csv_list(sequences) ? e2vq_io::files_from_csv(sequences[0], "TRAIN", cls, subdir, ".seq", nullptr, seq_files)
                                        : e2vq_io::resolve_filenames(sequences, ".seq", seq_files);
        if (rc || seq_files.empty()) { printf("%s\n", rc ? e2vq_last_error() : "No sequences given"); return 0; }
        printf("ECOZ2 C version: %s\n", ecoz2_version());
        printf("sequences: %zu\n", seq_files.size());
        printf("val_auto = %g\n", val_auto);
        ecoz2_set_random_seed(seed);
        auto ps = cptrs(seq_files);
        if (ecoz2_hmm_learn(N, type, ps.data(), (unsigned)ps.size(), epsilon, val_auto, max_iterations, ser ? 0 : 1, hmm_callback))
            printf("%s\n", e2vq_last_error());
        return 0;
    }
    if (cmd == "classify") {  // main_hmm_classify, src/hmm/mod.rs:219-283
        if (models.empty() || tt.empty() || (predictors.empty() == sequences.empty())) return usage();
        std::vector<std::string> hmm_files;
        e2vq_io::resolve_filenames(models, ".hmm", hmm_files);
        if (hmm_files.empty()) { printf("No models given\n"); return 0; }
        auto pm = cptrs(hmm_files);
        const char* c12n_file = c12n.empty() ? nullptr : c12n.c_str();
        if (!sequences.empty()) {
            std::vector<std::string> seq_files;
            int rc = is_csv_list(sequences) ? e2vq_io::files_from_csv(sequences[0], tt, cls, subdir, ".seq", nullptr, seq_files)
                                            : e2vq_io::resolve_filenames(sequences, ".seq", seq_files);
            if (rc) { printf("%s\n", e2vq_last_error()); return 0; }
            printf("ECOZ2 C version: %s\n", ecoz2_version());
            printf("number of HMM models: %zu  number of sequences: %zu\n", hmm_files.size(), seq_files.size());
            printf("show_ranked = %s\n", ranked ? "true" : "false");
            auto ps = cptrs(seq_files);
            if (ecoz2_hmm_classify(pm.data(), (unsigned)pm.size(), ps.data(), (unsigned)ps.size(), ranked, c12n_file))
                printf("%s\n", e2vq_last_error());
        } else {
            if (codebooks.empty()) return usage();
            std::vector<std::string> cb_files, prd_files;
            e2vq_io::resolve_filenames(codebooks, ".cbook", cb_files);
            if (cb_files.empty()) { printf("No codebooks given\n"); return 0; }
            int rc = is_csv_list(predictors) ? e2vq_io::files_from_csv(predictors[0], tt, cls, "", ".prd", &tmpl, prd_files)
                                             : e2vq_io::resolve_filenames(predictors, ".prd", prd_files);
            if (rc) { printf("%s\n", e2vq_last_error()); return 0; }
            auto pc = cptrs(cb_files), pp = cptrs(prd_files);
            if (ecoz2_hmm_classify_predictors(pm.data(), (unsigned)pm.size(), pc.data(), (int)pc.size(), pp.data(), (int)pp.size(),
                                              ranked, c12n_file))
                printf("%s\n", e2vq_last_error());
        }
        return 0;
    }
    return usage();
}

int main(int argc, char** argv)
{
    if (argc >= 2 && !strcmp(argv[1], "cversion")) {
        printf("%s\n", ecoz2_version());
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "seq") && !strcmp(argv[2], "show")) return seq_show(argc - 3, argv + 3);
    if (argc >= 3 && !strcmp(argv[1], "prd") && !strcmp(argv[2], "show")) return prd_show(argc - 3, argv + 3);
    if (argc >= 3 && !strcmp(argv[1], "hmm")) return hmm_cmd(argc - 2, argv + 2);
    if (argc >= 3 && !strcmp(argv[1], "nb")) return seq_model_cmd(true, argc - 2, argv + 2);
    if (argc >= 3 && !strcmp(argv[1], "mm")) return seq_model_cmd(false, argc - 2, argv + 2);
    if (argc < 3 || strcmp(argv[1], "vq") != 0) return usage();
    const std::string cmd = argv[2];
    if (cmd == "learn") return vq_learn(argc - 3, argv + 3);
    if (cmd == "quantize") return vq_quantize(argc - 3, argv + 3);
    if (cmd == "classify") return vq_classify(argc - 3, argv + 3);
    if (cmd == "show") return vq_show(argc - 3, argv + 3);
    return usage();
}
