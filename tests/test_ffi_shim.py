"""The reference's Rust call shape, restated in C (tests/ffi_shim.c): symbols declared without a return value, the learn calls
on a 128 MiB-stack helper thread, `#[repr(C)] Ecoz2ObserverRef` handed back through the callback, borrowed `*const c_char`
arrays that are never freed, the `-B` name without a terminator of its own (src/ecoz2_lib/mod.rs:51-70, 96-122, 271-342).
Built with gcc against libecoz2vq.so alone -- no HIP, no C++ runtime, none of this repo's headers on the caller's side."""
import json
import os
import subprocess

import numpy as np
import pytest

import ecoz2rs_amd as e

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CSRC = os.path.join(ROOT, "ecoz2rs_amd", "csrc")
P = 36


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    exe = tmp_path_factory.mktemp("shim") / "ffi_shim"
    subprocess.run(["gcc", "-O1", "-Wall", "-Wno-parentheses", "-o", str(exe), os.path.join(ROOT, "tests", "ffi_shim.c"),
                    "-L", CSRC, "-lecoz2vq", f"-Wl,-rpath,{CSRC}", "-lpthread"], check=True)
    return str(exe)


def test_shim_links_against_the_library_alone(shim):
    r = subprocess.run([shim, "version"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("VERSION ecoz2vq-mi355x")
    # every symbol the Rust extern block binds for this path resolves from the one shared object
    nm = subprocess.run(["nm", "-D", "--undefined-only", shim], capture_output=True, text=True).stdout
    assert all(f" {sym}" in nm for sym in ("ecoz2_version", "ecoz2_vq_learn", "ecoz2_vq_learn_using_base_codebook", "ecoz2_vq_quantize"))


def test_void_declared_entry_points_come_back_without_a_gpu(shim, tmp_path):
    """No device (this container) or a bad file: the library reports and RETURNS -- the Rust caller ignores the value, so
    nothing may abort or unwind through the C frames of the helper thread."""
    if e.lib.e2vq_device_count() > 0:
        pytest.skip("a HIP device is present: the GPU test below covers the call")
    f = tmp_path / "data" / "predictors" / "_" / "x.prd"
    e.formats.write_prd(str(f), "_", e.synth.synth_frames(1, 2, P, 0, 100))
    r = subprocess.run([shim, "learn", str(P), "0.05", "_", str(f)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ECOZ2_VQ_OUT_ROOT=str(tmp_path)))
    assert r.returncode == 0 and "DONE" in r.stdout and "STEP" not in r.stdout
    assert "no HIP device" in r.stderr or "no CPU path" in r.stderr


@pytest.mark.gpu
def test_config1_golden_files_through_the_rust_call_shape(shim, tmp_path):
    meta = json.load(open(os.path.join(GOLD, "config1.json")))
    frames = e.synth.synth_frames(meta["seed"], meta["classes"], P, 0, meta["T"])
    cuts = [0, 1, 1000, 1777, 4096, 9999, 10000]
    files = []
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        f = tmp_path / "data" / "predictors" / "_" / f"{i:05d}.prd"
        e.formats.write_prd(str(f), "_", frames[a:b])
        files.append(str(f))
    env = dict(os.environ, ECOZ2_VQ_OUT_ROOT=str(tmp_path), ECOZ2_VQ_MAX_CODEBOOK_SIZE=str(meta["max_M"]))
    r = subprocess.run([shim, "learn", str(P), repr(meta["eps"]), "_"] + files, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    steps = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("STEP ")]
    # the observer came back through the callback (ref_id read through the pointer), once per codebook size, same scalars
    assert [(int(s[1]), int(s[2])) for s in steps] == [(4242, g["M"]) for g in meta["levels"]]
    for s, g in zip(steps, meta["levels"]):
        assert [float.fromhex(x).hex() for x in s[3:6]] == [g["avg"], g["sigma"], g["inertia"]]
        name = f"eps_0.05_M_{g['M']:04d}.cbook"
        assert open(tmp_path / "data" / "codebooks" / "_" / name, "rb").read() == open(os.path.join(GOLD, "config1_" + name), "rb").read()
    # quantize, on the caller's own thread
    whole = tmp_path / "data" / "predictors" / "_" / "all.prd"
    e.formats.write_prd(str(whole), "_", frames)
    r = subprocess.run([shim, "quantize", str(tmp_path / "data" / "codebooks" / "_" / "eps_0.05_M_0016.cbook"), "1", str(whole)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    got = open(tmp_path / "data" / "sequences" / "M16" / "_" / "all.seq", "rb").read()
    assert got == open(os.path.join(GOLD, "config1_M0016.seq"), "rb").read()
    # -B with a name that carries no terminator of its own (mod.rs:295): trains 8 and 16 from the M = 4 codebook
    base = str(tmp_path / "data" / "codebooks" / "_" / "eps_0.05_M_0004.cbook")
    r = subprocess.run([shim, "learn_base", base, "1e9"] + files, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    steps = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("STEP ")]
    assert [int(s[2]) for s in steps] == [8, 16]
    from tests import oracle_lib

    oracle = oracle_lib.load()
    _c, _p, base_refl = e.formats.read_cbook(os.path.join(GOLD, "config1_eps_0.05_M_0004.cbook"))
    rc, levels_b, cbs_b = oracle.learn(frames, 1e9, meta["max_M"], base=base_refl)
    assert rc == 0 and [tuple(float.fromhex(x) for x in s[3:6]) for s in steps] == [tuple(c[1:4]) for c in cbs_b]
    for lv in levels_b:
        _c, _p, refl = e.formats.read_cbook(str(tmp_path / "data" / "codebooks" / "_" / f"eps_1e+09_M_{lv['M']:04d}.cbook"))
        assert np.array_equal(refl.view(np.uint64), lv["reflections"].view(np.uint64))
