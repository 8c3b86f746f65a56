"""CPU oracle of the `.seq` consumers (nb / mm / c12n) -- TEST INFRASTRUCTURE ONLY.

Restates, function by function, the reference's pure-Rust host code that IS present in /root/reference:
  nb    src/nb/nbayes.rs:14-153      mm    src/mm/markov.rs:16-167      c12n  src/c12n/mod.rs:8-236
  .seq  src/sequence/mod.rs:49-75    CBOR model files: utl::save_ser (src/utl/mod.rs:263-268) = serde_cbor 0.11.2
Only tests/ may import this module (the product is ecoz2rs_amd/csrc/seq_models.cpp).  Pure-Python loops: small
cases only.  Parity status: pinned by reference SOURCE TEXT (cited per function); the Rust toolchain is absent, so
no output of the reference itself could be generated here.  f32 arithmetic uses numpy.float32 scalars; log10 of an
f32 is libm's log10f (what Rust's f32::log10 lowers to on linux-gnu), called through ctypes.
"""
import ctypes
import math
import struct

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.log10f.restype = ctypes.c_float
_libm.log10f.argtypes = [ctypes.c_float]
f32 = np.float32


def log10f(x):
    return f32(_libm.log10f(float(x)))


# ---- sequence::load, src/sequence/mod.rs:49-75 -------------------------------------------------------------------
def load_seq(path):
    b = open(path, "rb").read()
    ident = b[:16].split(b"\0")[0].decode()
    if not ident.startswith("<sequence>"):  # :54-56
        raise ValueError("Not a sequence")
    class_name = b[16:112].split(b"\0")[0].decode()  # utl::read_class_name -> read_fixed_size_string
    length, = struct.unpack("<I", b[112:116])  # :60
    codebook_size, = struct.unpack("<I", b[116:120])  # :62
    symbols = list(struct.unpack("<%dH" % length, b[120:120 + 2 * length]))  # :66-68
    return dict(class_name=class_name, codebook_size=codebook_size, symbols=symbols)


# ---- nb: src/nb/nbayes.rs ------------------------------------------------------------------------------------------
def nb_learn(codebook_size, seq_filenames):
    """nbayes::learn :63-114"""
    seq = load_seq(seq_filenames[0])
    class_name = seq["class_name"]
    total_symbols = 0
    frequencies = [0] * seq["codebook_size"]
    for fn in seq_filenames:
        seq = load_seq(fn)
        if codebook_size != seq["codebook_size"]:
            raise ValueError(f"conformity error: codebook size: {codebook_size} != {seq['codebook_size']}")
        if class_name != seq["class_name"]:
            raise ValueError(f"conformity error: class_name: {class_name} != {seq['class_name']}")
        total_symbols += len(seq["symbols"])
        for s in seq["symbols"]:
            frequencies[s] += 1
    return dict(class_name=class_name, total_symbols=total_symbols, frequencies=frequencies)


def nb_prob_symbol(m, symbol):
    """:38-42 m-estimate"""
    return (float(m["frequencies"][symbol]) + 1.0) / float(m["total_symbols"] + len(m["frequencies"]))


def nb_log_prob_sequence(m, seq):
    """:45-54: fold(0.0, acc + log10(prob))"""
    acc = 0.0
    for s in seq["symbols"]:
        acc = acc + math.log10(nb_prob_symbol(m, s))
    return acc


# ---- mm: src/mm/markov.rs --------------------------------------------------------------------------------------------
EQ_EPSILON = f32(1e-5)  # :16


def ndarray_sum(xs):
    """ndarray 0.17 ArrayBase::sum of a contiguous slice = numeric_util::unrolled_fold (what the asserts evaluate)"""
    xs = [f32(x) for x in xs]
    acc, p = f32(0), [f32(0)] * 8
    while len(xs) >= 8:
        for k in range(8):
            p[k] = f32(p[k] + xs[k])
        xs = xs[8:]
    for k in range(4):
        acc = f32(acc + f32(p[k] + p[k + 4]))
    for x in xs[:7]:
        acc = f32(acc + x)
    return acc


def mm_learn(codebook_size, seq_filenames):
    """markov::learn :59-126 (f32 counters, add-one smoothing, row-stochastic asserts :117,122)"""
    seq = load_seq(seq_filenames[0])
    class_name = seq["class_name"]
    M = codebook_size
    pi = np.full(M, 1, dtype=np.float32)  # :70
    n_js = np.zeros(M, dtype=np.int32)  # :71
    a = np.full((M, M), 1, dtype=np.float32)  # :72
    for fn in seq_filenames:
        seq = load_seq(fn)
        if codebook_size != seq["codebook_size"]:
            raise ValueError("conformity error: codebook size")
        if class_name != seq["class_name"]:
            raise ValueError("conformity error: class_name")
        sy = seq["symbols"]
        pi[sy[0]] += f32(1)  # :102
        for j, k in zip(sy, sy[1:]):  # windows(2) :103-108
            n_js[j] += 1
            a[j, k] += f32(1)
    num_seqs = f32(len(seq_filenames))  # :112
    pi = pi / f32(num_seqs + f32(codebook_size))  # :115
    assert abs(f32(ndarray_sum(pi) - f32(1))) < EQ_EPSILON  # :117
    for j in range(M):  # :119-123
        a[j] = a[j] / f32(f32(n_js[j]) + f32(codebook_size))
        assert abs(f32(ndarray_sum(a[j]) - f32(1))) < EQ_EPSILON
    return dict(class_name=class_name, pi=pi, a=a)


def mm_log_prob_sequence(m, seq):
    """:43-49, f32 throughout"""
    sy = seq["symbols"]
    p = log10f(m["pi"][sy[0]])
    for t in range(len(sy) - 1):
        p = f32(p + log10f(m["a"][sy[t], sy[t + 1]]))
    return p


# ---- serde_cbor 0.11.2 encoding of the model structs -----------------------------------------------------------------
def _head(major, v):
    m = major << 5
    if v < 24:
        return bytes([m | v])
    if v <= 0xFF:
        return bytes([m | 24, v])
    if v <= 0xFFFF:
        return bytes([m | 25]) + struct.pack(">H", v)
    if v <= 0xFFFFFFFF:
        return bytes([m | 26]) + struct.pack(">I", v)
    return bytes([m | 27]) + struct.pack(">Q", v)


def _text(s):
    b = s.encode()
    return _head(3, len(b)) + b


def _f32(v):
    """serde_cbor Serializer::serialize_f32: f16 when `f32::from(f16::from_f32(v)) == v`, else the 4-byte form"""
    v = f32(v)
    if np.isinf(v):
        return b"\xf9\x7c\x00" if v > 0 else b"\xf9\xfc\x00"
    if np.isnan(v):
        return b"\xf9\x7e\x00"
    with np.errstate(over="ignore"):
        h = np.float16(v)
    if np.isfinite(h) and f32(h) == v:
        return b"\xf9" + struct.pack(">e", float(h))
    return b"\xfa" + struct.pack(">f", float(v))


def nb_cbor(m):
    """#[derive(Serialize)] struct NBayes {class_name, total_symbols, frequencies} as a 3-entry map with text keys"""
    out = _head(5, 3) + _text("class_name") + _text(m["class_name"]) + _text("total_symbols") + _head(0, m["total_symbols"])
    out += _text("frequencies") + _head(4, len(m["frequencies"])) + b"".join(_head(0, f) for f in m["frequencies"])
    return out


def _ndarray(arr):
    """ndarray's serde format (array_serde.rs): struct Array {v: 1u8, dim, data: sequence in logical order}"""
    out = _head(5, 3) + _text("v") + _head(0, 1) + _text("dim") + _head(4, arr.ndim) + b"".join(_head(0, d) for d in arr.shape)
    flat = arr.reshape(-1)
    return out + _text("data") + _head(4, flat.size) + b"".join(_f32(x) for x in flat)


def mm_cbor(m):
    return _head(5, 3) + _text("class_name") + _text(m["class_name"]) + _text("pi") + _ndarray(m["pi"]) + _text("a") + _ndarray(m["a"])


# ---- Rust formatting helpers -------------------------------------------------------------------------------------------
def rust_lower_exp(v):
    """`{:e}` of an f64: shortest digits, d.ddde<exp>"""
    if v == 0:
        return "-0e0" if math.copysign(1, v) < 0 else "0e0"
    mant, exp = ("%r" % abs(v)), 0
    s = np.format_float_scientific(abs(v), unique=True, trim="-", exp_digits=1)  # shortest digits
    mant, exp = s.split("e")
    return ("-" if v < 0 else "") + mant + "e" + str(int(exp))


def json_f32(v):
    """serde_json (ryu) of an f32 in the positional range"""
    return np.format_float_positional(f32(v), unique=True, trim="0")


# ---- c12n: src/c12n/mod.rs -------------------------------------------------------------------------------------------
class C12nResults:
    def __init__(self, model_class_names):  # :19-34
        n = len(model_class_names)
        self.model_class_names = list(model_class_names)
        self.result = [[0] * (n + 1) for _ in range(n + 1)]
        self.confusion = [[0] * (n + 1) for _ in range(n + 1)]
        self.y_true, self.y_pred = [], []
        self.out = []  # what the reference prints, piece by piece

    def add_case(self, class_id, seq_classname, probs, show_ranked, title):  # :36-104
        n = len(self.model_class_names)
        self.result[n][0] += 1
        self.result[class_id][0] += 1
        ranked = sorted(enumerate(probs), key=lambda t: t[1])  # stable, ascending (:52-53)
        predicted_id = ranked[n - 1][0]
        correct = class_id == predicted_id
        self.out.append("*" if correct else "_")
        self.y_true.append(seq_classname)
        self.y_pred.append(self.model_class_names[predicted_id])
        if show_ranked and not correct:
            self.out.append(title() + "\n")
            for index, r in enumerate(reversed(range(n))):
                model_id = ranked[r][0]
                model_class_name = self.model_class_names[r]  # (sic) :71
                mark = "*" if class_id == model_id else ""
                self.out.append("  [%2d] %-1s model: <%2d>  %s  : '%s'  r=%d\n" % (
                    index, mark, model_id, rust_lower_exp(ranked[model_id][1]), model_class_name, r))  # (sic) :77
                if class_id == model_id:
                    break
            self.out.append("\n")
        self.confusion[class_id][ranked[n - 1][0]] += 1
        if correct:
            self.result[n][1] += 1
            self.result[class_id][1] += 1
        else:
            for i in range(1, n):
                if ranked[n - 1 - i][0] == class_id:
                    self.result[n][i + 1] += 1
                    self.result[class_id][i + 1] += 1
                    break

    def report_results(self, class_names, out_base_name):  # :106-223
        n, o, res = len(self.model_class_names), self.out, self.result
        if res[n][0] == 0:
            return None
        margin = 0
        for i, name in list(enumerate(class_names))[:n]:
            if res[i][0] > 0:
                margin = max(margin, len(name.encode()))
        margin += 2
        pad = lambda s: s + " " * max(0, margin - len(s))
        o.append("\n\n")
        o.append(pad("") + " " + "Confusion matrix:\n")
        o.append(pad("") + " ")
        o.append("     ")
        for j in range(n):
            if res[j][0] > 0:
                o.append("%3d " % j)
        o.append("    tests   errors\n")
        for i, name in list(enumerate(class_names))[:n]:
            if res[i][0] == 0:
                continue
            o.append("\n")
            o.append(pad(name) + " ")
            o.append("%3d  " % i)
            num_errs = 0
            for j in range(n):
                if res[j][0] > 0:
                    o.append("%3d " % self.confusion[i][j])
                    if i != j:
                        num_errs += self.confusion[i][j]
            o.append("%8d%8d" % (res[i][0], num_errs))
        o.append("\n\n")
        o.append(pad("") + " " + "class     accuracy   tests       candidate order\n")
        num_classes, accuracy, avg_accuracy = 0, f32(0), f32(0)
        for class_id, name in list(enumerate(class_names))[:n + 1]:  # (the TOTAL branch is unreachable: n names)
            if res[class_id][0] == 0:
                continue
            num_tests, correct_tests = res[class_id][0], res[class_id][1]
            acc = f32(f32(correct_tests) / f32(num_tests))
            if class_id < n:
                num_classes += 1
                avg_accuracy = f32(avg_accuracy + acc)
                o.append(pad(name) + " ")
                o.append("  %3d    " % class_id)
            else:
                o.append("\n" + pad("") + " " + "  TOTAL  ")
                accuracy = acc
            o.append("  %6.2f%%    %4d       " % (float(f32(f32(100) * acc)), num_tests))
            for i in range(1, n + 1):
                o.append("%4d " % res[class_id][i])
            o.append("\n")
        accuracy = f32(accuracy * f32(100))
        avg_accuracy = f32(f32(avg_accuracy * f32(100)) / f32(num_classes))
        o.append("  avg_accuracy  %6.2f%%\n" % float(avg_accuracy))
        o.append("\n")
        summary = "{\n  \"accuracy\": %s,\n  \"avg_accuracy\": %s\n}" % (json_f32(accuracy), json_f32(avg_accuracy))
        o.append(f"{out_base_name}_classification.json saved\n")
        import json

        arr = lambda v: "[]" if not v else "[\n" + ",\n".join("    " + json.dumps(x, ensure_ascii=False) for x in v) + "\n  ]"
        true_pred = "{\n  \"y_true\": %s,\n  \"y_pred\": %s\n}" % (arr(self.y_true), arr(self.y_pred))
        o.append(f"{out_base_name}_y_true_pred.json saved\n")
        return summary, true_pred


def nb_classify(models, seq_filenames, show_ranked, codebook_size):
    """nbayes::classify :116-153; returns (stdout text, summary json, y_true_pred json, C12nResults)"""
    names = [m["class_name"] for m in models]
    c = C12nResults(names)
    c.out.append("Loading NBayes models\nClassifying sequences\n")
    for fn in seq_filenames:
        seq = load_seq(fn)
        if seq["class_name"] in names:
            probs = [nb_log_prob_sequence(m, seq) for m in models]
            c.add_case(names.index(seq["class_name"]), seq["class_name"], probs, show_ranked,
                       lambda: f"\n{fn}: '{seq['class_name']}'\n")
    c.out.append("\n")
    files = c.report_results(names, f"nb_{codebook_size}")
    return "".join(c.out), files, c


def mm_classify(models, seq_filenames, show_ranked, codebook_size):
    """markov::classify :128-167"""
    names = [m["class_name"] for m in models]
    c = C12nResults(names)
    c.out.append("Loading MM models\nClassifying sequences\n")
    for fn in seq_filenames:
        seq = load_seq(fn)
        if seq["class_name"] in names:
            probs = [float(mm_log_prob_sequence(m, seq)) for m in models]
            c.add_case(names.index(seq["class_name"]), seq["class_name"], probs, show_ranked,
                       lambda: f"\n{fn}: '{seq['class_name']}'")
    c.out.append("\n")
    files = c.report_results(names, f"mm_{codebook_size}")
    return "".join(c.out), files, c
