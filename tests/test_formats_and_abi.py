"""CPU tests: file formats (pinned .seq layout), C-ABI surface, host-side file resolution, CLI plumbing."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import ecoz2rs_amd as e

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_seq_layout_matches_reference_reader(tmp_path, oracle):
    """src/sequence/mod.rs:49-75 + src/utl/mod.rs:19-55: ident[16] class[96] u32 T, u32 M, T x u16, little endian."""
    sym = np.array([0, 1, 513, 65535, 7], dtype=np.uint16)
    p1, p2, p3 = (str(tmp_path / n) for n in ("a.seq", "b.seq", "c.seq"))
    e.check(e.lib.e2vq_seq_write(p1.encode(), b"Bm", 2048, sym.ctypes.data, len(sym)))  # product writer
    oracle.L.e2o_seq_save(p2.encode(), b"Bm", 2048, sym.ctypes.data, len(sym))          # oracle writer
    e.formats.write_seq(p3, "Bm", 2048, sym)                                            # python writer
    raw = open(p1, "rb").read()
    assert raw == open(p2, "rb").read() == open(p3, "rb").read()
    assert len(raw) == 16 + 96 + 8 + 2 * len(sym)
    assert raw[:10] == b"<sequence>" and raw[10:16] == b"\0" * 6
    assert raw[16:18] == b"Bm" and raw[18:112] == b"\0" * 94
    assert raw[112:116] == (5).to_bytes(4, "little") and raw[116:120] == (2048).to_bytes(4, "little")
    assert raw[120:122] == b"\x00\x00" and raw[124:126] == b"\x01\x02" and raw[126:128] == b"\xff\xff"
    cls, M, got = e.formats.read_seq(p1)
    assert (cls, M) == ("Bm", 2048) and np.array_equal(got, sym)
    with pytest.raises(ValueError):
        open(p3, "r+b").write(b"<sequenze>")
        e.formats.read_seq(p3)


def test_prd_and_cbook_roundtrip_across_writers(tmp_path, oracle):
    frames = e.synth.synth_frames(1, 2, 12, 0, 33)
    refl = np.random.default_rng(0).uniform(-0.5, 0.5, (8, 13))
    a, b = str(tmp_path / "x" / "a.prd"), str(tmp_path / "b.prd")
    e.check(e.lib.e2vq_prd_write(a.encode(), b"classA", 12, frames.ctypes.data, 33))
    oracle.L.e2o_prd_save(b.encode(), b"classA", 12, frames.ctypes.data, 33)
    assert open(a, "rb").read() == open(b, "rb").read()
    cls, P, got = e.formats.read_prd(a)
    assert (cls, P) == ("classA", 12) and np.array_equal(got, frames)
    name = (C.c_char * 96)()
    Pn, Tn = C.c_int(), C.c_int64()
    e.check(e.lib.e2vq_prd_info(a.encode(), name, C.byref(Pn), C.byref(Tn)))
    assert (name.value, Pn.value, Tn.value) == (b"classA", 12, 33)
    buf = np.zeros((33, 13))
    e.check(e.lib.e2vq_prd_read(a.encode(), buf.ctypes.data, 33))
    assert np.array_equal(buf, frames)
    c, d = str(tmp_path / "c.cbook"), str(tmp_path / "d.cbook")
    e.check(e.lib.e2vq_cbook_write(c.encode(), b"_", 12, 8, refl.ctypes.data))
    oracle.L.e2o_cbook_save(d.encode(), b"_", 12, 8, refl.ctypes.data)
    assert open(c, "rb").read() == open(d, "rb").read()
    assert np.array_equal(e.formats.read_cbook(c)[2], refl)
    # wrong ident / missing file fail loudly through the C-ABI
    assert e.lib.e2vq_cbook_info(a.encode(), name, C.byref(Pn), C.byref(Pn)) != 0
    assert "Not a codebook" in e.lib.e2vq_last_error().decode()
    assert e.lib.e2vq_prd_info(b"/nonexistent.prd", name, C.byref(Pn), C.byref(Tn)) != 0


def test_cabi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ecoz2_vq.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b((?:ecoz2|e2vq)_[a-z0-9_]+)\s*\(", header))
    declared -= {"ecoz2_vq_learn_callback_t"}
    assert {"ecoz2_version", "ecoz2_vq_learn", "ecoz2_vq_learn_using_base_codebook", "ecoz2_vq_quantize",
            "ecoz2_vq_show"} <= declared
    lib = C.CDLL(e.lib_path)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert b"ecoz2vq" in e.lib.ecoz2_version()


def test_no_cpu_fallback_without_device():
    """Without a HIP device every compute entry point must fail loudly (here: no GPU in the CPU test box)."""
    if e.lib.e2vq_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(e.Ecoz2Error, match="no HIP device"):
        e.VqSession(36)
    h = C.c_void_p()
    assert e.lib.e2vq_session_create(0, 36, C.byref(h)) != 0 and not h.value


def test_synth_frames_are_valid_and_counter_based(oracle):
    a = e.synth.synth_frames(20244, 20, 36, 1000, 300)
    b = e.synth.synth_frames(20244, 20, 36, 0, 2000)
    assert np.array_equal(a, b[1000:1300])  # any shard regenerates identical frames
    assert np.all(a[:, 0] >= 1.0) and np.all(np.abs(a[:, 1:]) <= a[:, :1])
    for r in a[:40]:
        st, pe, rc, _ = oracle.lpca_r(r, 36)
        assert st == 0 and abs(pe - 1.0) < 1e-8 and np.all(np.abs(rc[1:]) <= 0.9500001)


def _cli(*args, cwd):
    exe = os.path.join(ROOT, "ecoz2rs_amd", "csrc", "ecoz2")
    return subprocess.run([exe, *args], cwd=cwd, capture_output=True, text=True, timeout=120)


def test_cli_file_resolution_mirrors_reference(tmp_path):
    """utl::resolve_files3 / get_files_from_csv (src/utl/mod.rs:112-193) and the mains' messages (src/vq/mod.rs)."""
    frames = e.synth.synth_frames(2, 2, 36, 0, 10)
    for cls, sel in (("A", "00002"), ("A", "00001"), ("B", "00003")):
        e.formats.write_prd(str(tmp_path / "data" / "predictors" / cls / f"{sel}.prd"), cls, frames)
    (tmp_path / "data" / "predictors" / "A" / "notes.txt").write_text("x")
    e.formats.write_cbook(str(tmp_path / "cb.cbook"), "_", np.zeros((2, 37)))
    (tmp_path / "tt.csv").write_text("# comment\ntt,class,selection\nTRAIN,A,00001\nTEST,A,00002\nTRAIN,B,00003\n")
    r = _cli("vq", "quantize", "--codebook", "cb.cbook", "--predictors", "data/predictors", cwd=tmp_path)
    assert "number of predictor files: 3" in r.stdout and "nom_raas = cb.cbook" in r.stdout
    r = _cli("vq", "quantize", "--codebook", "cb.cbook", "--predictors", "tt.csv", "--tt", "TRAIN",
             "--predictors-dir-template", "data/predictors/{class}/{selection}.prd", cwd=tmp_path)
    assert "number of predictor files: 2" in r.stdout
    r = _cli("vq", "quantize", "--codebook", "cb.cbook", "--predictors", "tt.csv", "--tt", "TRAIN", "--class-name", "B",
             "--predictors-dir-template", "data/predictors/{class}/{selection}.prd", cwd=tmp_path)
    assert "number of predictor files: 1" in r.stdout
    r = _cli("vq", "learn", "-B", "cb.cbook", "-P", "36", "--predictors", "data/predictors", cwd=tmp_path)
    assert "Only one of base codebook or prediction order expected" in r.stdout and r.returncode == 0
    r = _cli("vq", "learn", "-P", "36", "--predictors", "tt.csv", cwd=tmp_path)
    assert "predictor_filenames: 2" in r.stdout and "epsilon=0.05" in r.stdout and "codebook_class_name=_" in r.stdout
    r = _cli("vq", "show", "cb.cbook", cwd=tmp_path)
    assert "className='_', M=2, P=36" in r.stdout
    assert "ecoz2vq" in _cli("cversion", cwd=tmp_path).stdout


def test_cli_seq_show_and_prd_show(tmp_path):
    """Sequence::show output format (src/sequence/mod.rs:17-47); prd show header as in notes.md:77-85."""
    e.formats.write_seq(str(tmp_path / "a.seq"), "Bm", 256, np.arange(45, dtype=np.uint16))
    e.formats.write_seq(str(tmp_path / "b.seq"), "C", 4, np.array([3, 1, 2], dtype=np.uint16))
    r = _cli("seq", "show", "a.seq", "b.seq", cwd=tmp_path)
    lines = r.stdout.splitlines()
    assert lines[0] == "<Bm(M=256,L=45): 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, ..., 35, 36, 37, 38, 39, 40, 41, 42, 43, 44>"
    assert lines[1] == "<C(M=4,L=3): 3, 1, 2>"
    assert _cli("seq", "show", "-L", "a.seq", cwd=tmp_path).stdout.strip() == "45"
    assert "44>" in _cli("seq", "show", "--full", "a.seq", cwd=tmp_path).stdout and "..." not in _cli(
        "seq", "show", "--full", "a.seq", cwd=tmp_path).stdout
    fr = e.synth.synth_frames(1, 1, 36, 0, 3)
    e.formats.write_prd(str(tmp_path / "x.prd"), "HB", fr)
    out = _cli("prd", "show", "--from", "0", "--to", "2", "x.prd", cwd=tmp_path).stdout.splitlines()
    assert out[1] == "# className='HB', T=3, P=36" and out[2] == "r0,r1,r2"
    assert out[3] == ",".join(f"{v:.5f}" for v in fr[0, :3])
    # defaults of the reference's options (src/prd/mod.rs:45-51): --from 1, --to 0 = up to P
    out = _cli("prd", "show", "x.prd", cwd=tmp_path).stdout.splitlines()
    assert out[2] == ",".join(f"r{n}" for n in range(1, 37)) and len(out) == 6
    # -k: reflection coefficients of each vector = lpca_r (src/lpc/lpca_r_rs.rs:8-43) on its autocorrelation
    from tests import oracle_lib

    oracle = oracle_lib.load()
    out = _cli("prd", "show", "-k", "-f", "1", "-t", "5", "x.prd", cwd=tmp_path).stdout.splitlines()
    assert out[2] == "k1,k2,k3,k4,k5"
    for t in range(3):
        st, _pe, rc, _a = oracle.lpca_r(fr[t], 36)
        assert st == 0 and out[3 + t] == ",".join(f"{v:.5f}" for v in rc[1:6])


def test_prd_show_file_symbol_is_what_the_reference_binds(tmp_path, capfd):
    """ecoz2_prd_show_file(prd_filename, show_reflections, from, to): src/ecoz2_lib/mod.rs:89-94, called on the main
    thread by prd::prd_show_file (:220-239)."""
    fr = e.synth.synth_frames(3, 1, 12, 0, 5)
    e.formats.write_prd(str(tmp_path / "y.prd"), "W", fr)
    e.vq.prd_show_file(tmp_path / "y.prd", False, 0, 0)
    out = capfd.readouterr().out.splitlines()
    assert out[1] == "# className='W', T=5, P=12" and out[2] == ",".join(f"r{n}" for n in range(13))
    assert out[3 + 4] == ",".join(f"{v:.5f}" for v in fr[4])
    assert e.lib.ecoz2_prd_show_file(str(tmp_path / "missing.prd").encode(), 0, 1, 0) != 0


def test_cli_seq_show_pickle(tmp_path):
    """`seq show --pickle` (src/seq/mod.rs:88-118): list of symbol lists, resolved from a tt-list; read back with pickle."""
    import pickle

    rng = np.random.default_rng(0)
    want = []
    rows = ["tt,class,selection"]
    for cls, sel, n in (("A", "00001", 5), ("A", "00002", 300), ("B", "00003", 70000)):
        sym = rng.integers(0, 2048, n).astype(np.uint16)
        e.formats.write_seq(str(tmp_path / "data" / "sequences" / "M2048" / cls / f"{sel}.seq"), cls, 2048, sym)
        rows.append(f"TRAIN,{cls},{sel}")
        want.append(sym.tolist())
    (tmp_path / "tt.csv").write_text("\n".join(rows) + "\n")
    r = _cli("seq", "show", "--pickle", "out.pkl", "-M", "2048", "--tt", "TRAIN", "tt.csv", cwd=tmp_path)
    assert '3 sequence(s) saved to "out.pkl"' in r.stdout
    assert pickle.load(open(tmp_path / "out.pkl", "rb")) == want
    r = _cli("seq", "show", "--pickle", "a.pkl", "-M", "2048", "--tt", "TRAIN", "--class-name", "A", "tt.csv", cwd=tmp_path)
    assert pickle.load(open(tmp_path / "a.pkl", "rb")) == want[:2]
    r = _cli("seq", "show", "--pickle", "x.pkl", "tt.csv", cwd=tmp_path)
    assert "--codebook-size and --tt required when --pickle given" in r.stdout
    # directories are walked and the files sorted as Vec<PathBuf>::sort does (src/utl/mod.rs:216-218): component by
    # component, so "d/a/x.seq" comes before "d/a-b/y.seq" although '-' < '/' as bytes
    order = {"d/a/x.seq": [1], "d/a-b/y.seq": [2], "d/a/b/z.seq": [3], "d/a0.seq": [4]}
    for name, sym in order.items():
        e.formats.write_seq(str(tmp_path / name), "A", 2048, np.array(sym, dtype=np.uint16))
    r = _cli("seq", "show", "--pickle", "o.pkl", "-M", "2048", "--tt", "TRAIN", "d", cwd=tmp_path)
    assert pickle.load(open(tmp_path / "o.pkl", "rb")) == [[3], [1], [2], [4]]  # a/b/z < a/x < a-b/y < a0.seq


def test_prd_cbook_header_field_order_is_resolved_by_payload_size(tmp_path):
    """Only .seq is pinned by the reference; for .prd/.cbook the reader accepts either order of the two u32 header
    fields (decided by the payload size) and rejects anything inconsistent."""
    import struct

    frames = e.synth.synth_frames(1, 1, 12, 0, 40)
    hdr = lambda ident, cls: ident.encode().ljust(16, b"\0") + cls.encode().ljust(96, b"\0")
    swapped = tmp_path / "swapped.prd"
    swapped.write_bytes(hdr("<predictor>", "Z") + struct.pack("<II", 12, 40) + frames.tobytes())  # (P, T) order
    name = (C.c_char * 96)()
    Pn, Tn = C.c_int(), C.c_int64()
    e.check(e.lib.e2vq_prd_info(str(swapped).encode(), name, C.byref(Pn), C.byref(Tn)))
    assert (Pn.value, Tn.value) == (12, 40)
    buf = np.zeros((40, 13))
    e.check(e.lib.e2vq_prd_read(str(swapped).encode(), buf.ctypes.data, 40))
    assert np.array_equal(buf, frames)
    bad = tmp_path / "bad.prd"
    bad.write_bytes(hdr("<predictor>", "Z") + struct.pack("<II", 41, 12) + frames.tobytes())
    assert e.lib.e2vq_prd_info(str(bad).encode(), name, C.byref(Pn), C.byref(Tn)) != 0
    assert "do not match the payload size" in e.lib.e2vq_last_error().decode()
    refl = np.random.default_rng(1).uniform(-0.4, 0.4, (8, 13))
    cb = tmp_path / "swapped.cbook"
    cb.write_bytes(hdr("<codebook>", "_") + struct.pack("<II", 8, 12) + refl.tobytes())  # (M, P) order
    Mn = C.c_int()
    e.check(e.lib.e2vq_cbook_info(str(cb).encode(), name, C.byref(Pn), C.byref(Mn)))
    assert (Pn.value, Mn.value) == (12, 8)
