"""GPU parity of the HMM consumers: the HIP kernels, called through the C-ABI, against the CPU
oracle -- bit-exact (mantissa / exponent of P(O), every accumulator word, every re-estimated parameter) -- and
config 5 of BASELINE.json end to end on one GPU: predictors -> vq learn M=1024 -> vq quantize -> hmm learn ->
hmm classify (sequences and predictors+codebooks), with nb / mm as the cross-check the reference's Rust pins."""
import os
import subprocess

import numpy as np
import pytest

import ecoz2rs_amd as e
from tests import oracle_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 36


@pytest.fixture(scope="module")
def H():
    return oracle_lib.load_hmm()


def _ragged(rng, M, lens):
    return [rng.integers(0, M, n).astype(np.uint16) for n in lens]


@pytest.mark.parametrize("N,M,typ", [(1, 8, 0), (2, 16, 0), (5, 256, 3), (5, 1024, 2), (17, 64, 0), (64, 128, 3), (64, 32, 0),
                                     (65, 32, 0), (100, 64, 3), (130, 16, 2), (257, 8, 0)])  # > 64 states: a workgroup per sequence
def test_score_bit_exact(H, N, M, typ):
    """scaled forward pass: P(O) = mant * 2^exp2 of every (sequence, model) pair equals the oracle's, word for word"""
    H.seed(100 + N)
    rng = np.random.default_rng(N * 1000 + M)
    models = [H.init(N, M, typ), H.init(N, M, 0), H.init(max(1, N // 2), M, 1)]
    seqs = _ragged(rng, M, [1, 2, 63, 64, 65, 128, 129, 300, 7, 1000]) + [np.zeros(0, dtype=np.uint16)]
    got = e.hmm.score(models, seqs)
    for s, sq in enumerate(seqs):
        for k, (pi, A, B) in enumerate(models):
            st, m, ex = H.forward(pi, A, B, sq)
            assert got["status"][s, k] == st
            assert got["mant"][s, k] == m and got["exp2"][s, k] == ex, (s, k)
            assert got["log_prob"][s, k] == H.log_prob(m, ex)
    assert got["log_prob"][-1].tolist() == [0.0, 0.0, 0.0]  # the empty sequence: P = 1


def test_score_status_codes(H):
    H.seed(5)
    pi, A, B = H.init(4, 8, 3)
    B0 = B.copy()
    B0[:, 5] = 0.0
    seqs = [np.array([1, 5, 2], dtype=np.uint16), np.array([1, 2, 3], dtype=np.uint16), np.array([1, 9], dtype=np.uint16)]
    got = e.hmm.score([(pi, A, B0)], seqs)
    assert got["status"][:, 0].tolist() == [1, 0, 2]
    assert got["log_prob"][0, 0] == -np.inf and got["log_prob"][2, 0] == -np.inf and np.isfinite(got["log_prob"][1, 0])


@pytest.mark.parametrize("N,M,typ", [(1, 8, 0), (3, 16, 3), (5, 64, 3), (5, 1024, 2), (16, 32, 0), (64, 16, 3), (65, 16, 0), (96, 32, 3),
                                     (130, 8, 2)])
def test_estep_accumulators_bit_exact(H, N, M, typ):
    """Baum-Welch E-step: the exact fixed-point expected counts (PI, AN, AD, BN, BD, used / skipped) equal the oracle's"""
    H.seed(200 + N)
    rng = np.random.default_rng(N + M)
    pi, A, B = H.init(N, M, typ)
    seqs = _ragged(rng, M, [1, 2, 5, 64, 65, 200, 33, 90, 17, 128])
    if typ == 3 and N >= 3:
        B = B.copy()
        B[:, 0] = 0.0  # sequences containing symbol 0 cannot be emitted: skipped, and counted as such
        B /= B.sum(1, keepdims=True)
    acc_o, res = H.accumulate(pi, A, B, seqs)
    acc, mant, ex, st = e.hmm.estep(pi, A, B, seqs)
    assert st.tolist() == [r[0] for r in res]
    assert np.array_equal(acc, acc_o)
    for s, r in enumerate(res):
        if r[0] == 0:
            assert mant[s] == r[1] and ex[s] == r[2]


@pytest.mark.parametrize("N,M,typ,eps,auto,maxit", [(5, 32, 3, 1e-5, 0.3, -1), (3, 16, 0, 0.0, 0.05, -1), (8, 64, 2, 1e-4, 0.0, 6),
                                                     (64, 16, 3, 1e-5, 0.3, 3), (80, 16, 3, 1e-5, 0.3, 3), (70, 8, 0, 0.0, 0.05, 2)])
def test_training_bit_exact(H, N, M, typ, eps, auto, maxit):
    """whole Baum-Welch (E-steps, M-steps, epsilon restriction, stopping rule): parameters and the measure per
    iteration equal the oracle's bit for bit"""
    H.seed(300 + N)
    rng = np.random.default_rng(7 * N + M)
    pi, A, B = H.init(N, M, typ)
    # sequences with some structure: symbols drift upwards along the sequence
    seqs = []
    for _ in range(40):
        T = int(rng.integers(20, 120))
        seqs.append(np.clip((np.linspace(0, M - 1, T) + rng.normal(0, M / 8, T)).round(), 0, M - 1).astype(np.uint16))
    po, Ao, Bo, hist_o = H.learn(pi, A, B, seqs, eps, auto, maxit)
    pg, Ag, Bg, hist = e.hmm.train(pi, A, B, seqs, eps, auto, maxit)
    assert hist == hist_o and (maxit < 0 or len(hist) <= maxit)
    for a, b in ((po, pg), (Ao, Ag), (Bo, Bg)):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def _corpus(tmp_path, n_classes, n_train, n_test, seed=77, phones=14, string_len=6):
    """synthetic 20-class corpus (config 5; the LPC front-end is CPU work outside this path, so the predictor vectors
    are synthesised directly): a class is a string of `phones` (prototype spectra); a recording = its phones in order,
    each held for 12-30 frames drawn around the phone's prototype (e2vq_synth_frames with one class = one prototype)."""
    rng = np.random.default_rng(seed)
    strings = []
    while len(strings) < n_classes:
        s = tuple(rng.choice(phones, string_len, replace=False))
        if s not in strings:
            strings.append(s)
    rows, files = ["tt,class,selection"], {}
    for c, string in enumerate(strings):
        cls = f"C{c:02d}"
        for k in range(n_train + n_test):
            segs = [e.synth.synth_frames(9000 + int(ph), 1, P, int(rng.integers(0, 1 << 20)), int(rng.integers(12, 30)))
                    for ph in string]
            f = tmp_path / "data" / "predictors" / cls / f"{k:05d}.prd"
            e.formats.write_prd(str(f), cls, np.concatenate(segs))
            tt = "TRAIN" if k < n_train else "TEST"
            rows.append(f"{tt},{cls},{k:05d}")
            files.setdefault((cls, tt), []).append(str(f))
    (tmp_path / "tt.csv").write_text("\n".join(rows) + "\n")
    return [f"C{c:02d}" for c in range(n_classes)], files


def test_config5_end_to_end_20_classes(H, tmp_path, monkeypatch, capfd):
    """BASELINE.json configs[4] on one GPU: predictors -> vq learn M=1024 -> vq quantize -> hmm learn (per class) ->
    hmm classify on the TEST recordings, from sequences and from predictors+codebook (must agree exactly), plus the nb /
    mm classifiers over the same .seq files.  Checks: the .hmm files equal the oracle's training on the same
    sequences; the c12n CSV; >= 90 % of the TEST recordings classified correctly by every classifier."""
    monkeypatch.setenv("ECOZ2_VQ_OUT_ROOT", str(tmp_path))
    monkeypatch.setenv("ECOZ2_VQ_QUIET", "1")
    monkeypatch.setenv("NO_COLOR", "1")
    monkeypatch.setenv("ECOZ2_VQ_MAX_CODEBOOK_SIZE", "1024")
    monkeypatch.chdir(tmp_path)
    M, N = 1024, 6
    classes, files = _corpus(tmp_path, 20, 12, 5)
    train_prd = sorted(sum((files[(c, "TRAIN")] for c in classes), []))
    test_prd = sorted(sum((files[(c, "TEST")] for c in classes), []))
    # vq learn on all TRAIN recordings (one codebook, class "_"), then quantize everything
    e.vq_learn(None, P, 0.05, "_", train_prd)
    cbook = str(tmp_path / "data" / "codebooks" / "_" / "eps_0.05_M_1024.cbook")
    assert os.path.exists(cbook)
    e.vq_quantize(cbook, train_prd + test_prd)
    seq_of = lambda prd: prd.replace("/predictors/", f"/sequences/M{M}/").replace(".prd", ".seq")
    assert all(os.path.exists(seq_of(f)) for f in train_prd + test_prd)
    # hmm learn per class (cascade-3, seeded), checked against the oracle on the same sequences
    hmm_dir = tmp_path / "data" / "hmms" / f"N{N}__M{M}_t3__a0.3"
    for ci, cls in enumerate(classes):
        seqs = [seq_of(f) for f in files[(cls, "TRAIN")]]
        e.hmm.set_random_seed(1000 + ci)
        seen = []
        e.hmm.hmm_learn(N, 3, seqs, 1e-5, 0.3, -1, callback=lambda v, x: seen.append((v, x)))
        cls_r, pi, A, B = e.hmm.load_model(hmm_dir / f"{cls}.hmm")
        assert cls_r == cls and seen and all(v == "sum_log_prob" for v, _ in seen)
        if ci < 3:  # the oracle's training on the same symbols, same seed: identical model and measure
            H.seed(1000 + ci)
            pi0, A0, B0 = H.init(N, M, 3)
            po, Ao, Bo, hist = H.learn(pi0, A0, B0, [e.formats.read_seq(s)[2] for s in seqs], 1e-5, 0.3, -1)
            assert [x for _v, x in seen] == hist
            assert np.array_equal(B.view(np.uint64), Bo.view(np.uint64)) and np.array_equal(A.view(np.uint64), Ao.view(np.uint64))
            csv = open(hmm_dir / f"{cls}.csv").read().splitlines()
            assert csv[1] == "I,sum_log_prob" and len(csv) == 2 + len(hist) and float(csv[-1].split(",")[1]) == hist[-1]
    models = sorted(str(p) for p in hmm_dir.glob("*.hmm"))
    assert len(models) == 20
    test_seq = [seq_of(f) for f in test_prd]
    capfd.readouterr()
    e.hmm.hmm_classify_sequences(models, test_seq, True, str(tmp_path / "c12n_seq.csv"))
    out_seq = capfd.readouterr().out
    e.hmm.hmm_classify_predictors(models, [cbook], test_prd, True, str(tmp_path / "c12n_prd.csv"))
    out_prd = capfd.readouterr().out
    # the classification CSV of `--c12n` (CHANGELOG.md:273-284)
    rows = open(tmp_path / "c12n_seq.csv").read().splitlines()
    assert rows[0] == f"# num_models=20  M={M}  num_seqs=100" and rows[1] == "seq_filename,seq_class_name,correct,rank"
    assert len(rows) == 102 and all(r.split(",")[2] in "*!" and int(r.split(",")[3]) >= 1 for r in rows[2:])
    correct = sum(r.split(",")[2] == "*" for r in rows[2:])
    assert correct >= 90, correct
    assert f"{correct / 100 * 100:6.2f}%" in out_seq.split("TOTAL")[1]  # the C report does reach its TOTAL row
    # quantising on the fly gives the same symbols, hence the same scores, ranks and report
    rows_p = open(tmp_path / "c12n_prd.csv").read().splitlines()
    assert [r.split(",")[1:] for r in rows_p[2:]] == [r.split(",")[1:] for r in rows[2:]]
    tail = lambda out: out.split("Confusion matrix:")[1].split("c12n_")[0]
    assert tail(out_prd) == tail(out_seq)
    # the scores themselves against the oracle, for a few recordings
    loaded = [e.hmm.load_model(m) for m in models]
    got = e.hmm.score([m[1:] for m in loaded], [e.formats.read_seq(s)[2] for s in test_seq[:6]])
    for s in range(6):
        sy = e.formats.read_seq(test_seq[s])[2]
        for k in (0, 7, 19):
            st, m, ex = H.forward(*loaded[k][1:], sy)
            assert (got["status"][s, k], got["mant"][s, k], got["exp2"][s, k]) == (st, m, ex)
    # nb / mm over the same .seq files (the consumers whose arithmetic the reference's Rust pins)
    for kind in ("nb", "mm"):
        learn, classify = getattr(e.classify, kind + "_learn"), getattr(e.classify, kind + "_classify")
        mfiles = [learn(M, [seq_of(f) for f in files[(cls, "TRAIN")]]) for cls in classes]
        capfd.readouterr()
        classify(mfiles, test_seq, False, M)
        capfd.readouterr()
        import json
        tp = json.load(open(tmp_path / f"{kind}_{M}_y_true_pred.json"))
        acc = np.mean([a == b for a, b in zip(tp["y_true"], tp["y_pred"])])
        # (mm at M = 1024 smooths every one of its 1024 x 1024 transitions with add-one counts, markov.rs:70-74: with
        # ~1 400 training symbols per class it is far weaker than nb / hmm -- chance is 0.05)
        assert acc >= (0.9 if kind == "nb" else 0.3), (kind, acc)


@pytest.mark.parametrize("workers", [2, 3])
def test_hmm_workers_give_identical_models_and_reports(tmp_path, monkeypatch, capfd, workers):
    """ECOZ2_VQ_GPUS=N behind ecoz2_hmm_learn / ecoz2_hmm_classify / ecoz2_hmm_classify_predictors: the
    sequences (predictor files) are dealt to N workers -- sharing the one GPU here --; the E-step's expected counts are
    exact int64 limb sums, added up over the workers, so the .hmm bytes, the training measure, the c12n CSV and the
    report are those of a single worker, byte for byte."""
    monkeypatch.setenv("ECOZ2_VQ_OUT_ROOT", str(tmp_path))
    monkeypatch.setenv("ECOZ2_VQ_QUIET", "1")
    monkeypatch.setenv("NO_COLOR", "1")
    monkeypatch.setenv("ECOZ2_VQ_MAX_CODEBOOK_SIZE", "64")
    monkeypatch.chdir(tmp_path)
    M, N = 64, 5
    classes, files = _corpus(tmp_path, 4, 7, 4, seed=11)
    train_prd = sorted(sum((files[(c, "TRAIN")] for c in classes), []))
    test_prd = sorted(sum((files[(c, "TEST")] for c in classes), []))
    monkeypatch.setenv("ECOZ2_VQ_GPUS", "1")
    e.vq_learn(None, P, 0.05, "_", train_prd)
    cbook = str(tmp_path / "data" / "codebooks" / "_" / f"eps_0.05_M_{M:04d}.cbook")
    e.vq_quantize(cbook, train_prd + test_prd)
    seq_of = lambda prd: prd.replace("/predictors/", f"/sequences/M{M}/").replace(".prd", ".seq")
    hmm_dir = tmp_path / "data" / "hmms" / f"N{N}__M{M}_t3__a0.05"

    def run(nw):
        monkeypatch.setenv("ECOZ2_VQ_GPUS", str(nw))
        hists, blobs = [], []
        for ci, cls in enumerate(classes):
            e.hmm.set_random_seed(77 + ci)
            seen = []
            e.hmm.hmm_learn(N, 3, [seq_of(f) for f in files[(cls, "TRAIN")]], 1e-5, 0.05, -1,
                            callback=lambda v, x: seen.append(x))
            hists.append(seen)
            blobs.append(open(hmm_dir / f"{cls}.hmm", "rb").read())
        models = sorted(str(p) for p in hmm_dir.glob("*.hmm"))
        capfd.readouterr()
        e.hmm.hmm_classify_sequences(models, [seq_of(f) for f in test_prd], True, str(tmp_path / f"s{nw}.csv"))
        out_s = capfd.readouterr().out
        e.hmm.hmm_classify_predictors(models, [cbook], test_prd, True, str(tmp_path / f"p{nw}.csv"))
        out_p = capfd.readouterr().out
        strip = lambda o: o.split("Confusion matrix:")[1].split(".csv saved")[0].rsplit("\n", 1)[0]
        return hists, blobs, open(tmp_path / f"s{nw}.csv").read(), open(tmp_path / f"p{nw}.csv").read(), strip(out_s), strip(out_p)

    one = run(1)
    many = run(workers)
    assert many[0] == one[0], "training measure differs"
    assert many[1] == one[1], ".hmm bytes differ"
    assert many[2] == one[2] and many[3] == one[3] and many[4] == one[4] and many[5] == one[5]
    assert len(one[0][0]) >= 2  # (several E-steps ran)
    # bounded memory (round 4): ecoz2_hmm_classify_predictors streams the corpus in units of whole files through fixed
    # staging buffers -- with a chunk far below the corpus (units of a few files), and below a single file's length
    # (the file is streamed piece by piece into the symbol buffer), CSV and report stay byte-identical
    models = sorted(str(p) for p in hmm_dir.glob("*.hmm"))
    strip = lambda o: o.split("Confusion matrix:")[1].split(".csv saved")[0].rsplit("\n", 1)[0]
    lengths = [e.formats.read_prd(f)[2].shape[0] for f in test_prd]
    for chunk in (3 * max(lengths), 64):
        assert chunk < sum(lengths) and (chunk != 64 or min(lengths) > 64)
        monkeypatch.setenv("ECOZ2_VQ_CLASSIFY_CHUNK", str(chunk))
        for nw in (1, workers):
            monkeypatch.setenv("ECOZ2_VQ_GPUS", str(nw))
            capfd.readouterr()
            e.hmm.hmm_classify_predictors(models, [cbook], test_prd, True, str(tmp_path / f"pc{chunk}_{nw}.csv"))
            out_p = capfd.readouterr().out
            assert open(tmp_path / f"pc{chunk}_{nw}.csv").read() == one[3] and strip(out_p) == one[5], (chunk, nw)


def test_hmm_cli_end_to_end(tmp_path):
    """`ecoz2 hmm learn / classify / show` through the CLI binary with the reference's flags (src/hmm/mod.rs:40-145)"""
    exe = os.path.join(ROOT, "ecoz2rs_amd", "csrc", "ecoz2")
    env = dict(os.environ, NO_COLOR="1", ECOZ2_VQ_MAX_CODEBOOK_SIZE="32")
    for k in ("ECOZ2_VQ_OUT_ROOT", "ECOZ2_VQ_QUIET"):
        env.pop(k, None)

    def run(*args):
        r = subprocess.run([exe, *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        return r.stdout

    classes, _files = _corpus(tmp_path, 3, 6, 3, seed=5, phones=8, string_len=4)
    run("vq", "learn", "-P", "36", "--predictors", "tt.csv")
    run("vq", "quantize", "--codebook", "data/codebooks/_/eps_0.05_M_0032.cbook", "--predictors", "data/predictors")
    for cls in classes:
        out = run("hmm", "learn", "-N", "4", "-M", "32", "-s", "3", "-I", "8", "--class-name", cls, "--sequences", "tt.csv")
        assert "sequences: 6" in out and "val_auto = 0.3" in out and f"class '{cls}'  N=4 M=32 type=3" in out
        assert os.path.exists(tmp_path / "data" / "hmms" / "N4__M32_t3__a0.3_I8" / f"{cls}.hmm")
    out = run("hmm", "classify", "--models", "data/hmms/N4__M32_t3__a0.3_I8", "--tt", "TEST", "-M", "32", "-r",
              "--c12n", "c12n.csv", "--sequences", "tt.csv")
    assert "number of HMM models: 3  number of sequences: 9" in out and "Confusion matrix:" in out and "TOTAL" in out
    rows = open(tmp_path / "c12n.csv").read().splitlines()
    assert rows[0] == "# num_models=3  M=32  num_seqs=9" and rows[2].startswith("data/sequences/M32/C00/00006.seq,C00,")
    out2 = run("hmm", "classify", "--models", "data/hmms/N4__M32_t3__a0.3_I8", "--tt", "TEST", "-M", "32",
               "--predictors", "tt.csv", "--predictors-dir-template", "data/predictors/{class}/{selection}.prd",
               "--codebooks", "data/codebooks/_/eps_0.05_M_0032.cbook")
    assert out2.split("Confusion matrix:")[1] == out.split("Confusion matrix:")[1].split("c12n.csv saved")[0]
    out = run("hmm", "show", "--hmm", "data/hmms/N4__M32_t3__a0.3_I8/C01.hmm")
    assert "className='C01', N=4, M=32" in out and out.count(" [3]: ") == 2
    # more states than a wavefront has lanes (the reference's -N is free): same flags, the workgroup-per-sequence kernels
    for cls in classes:
        out = run("hmm", "learn", "-N", "70", "-M", "32", "-s", "3", "-I", "2", "--class-name", cls, "--sequences", "tt.csv")
        assert f"class '{cls}'  N=70 M=32 type=3" in out
    out = run("hmm", "classify", "--models", "data/hmms/N70__M32_t3__a0.3_I2", "--tt", "TEST", "-M", "32", "--sequences", "tt.csv")
    assert "number of HMM models: 3  number of sequences: 9" in out and "Confusion matrix:" in out
