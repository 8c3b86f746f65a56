#!/usr/bin/env python3
"""Regenerates the golden fixtures of config 1 (BASELINE.json configs[0]: vq learn on 10k synthetic
P=36 frames, M=16) with the CPU oracle.  Run from the repo root:  python tests/golden/make_golden.py

Outputs (tests/golden/):
  config1.json                      seed, sizes, sha256 of the synthetic frames, per-level scalars (hex floats)
  config1_eps_0.05_M_00NN.cbook     the oracle's codebooks, M = 2, 4, 8, 16
  config1_M0016.seq                 the oracle's symbols for all 10k frames against the M=16 codebook
signal_frame.inputs is the reference's own LPC fixture (/root/reference/signal_frame.inputs, data only).
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

SEED, CLASSES, P, T, MAX_M, EPS = 20241, 4, 36, 10000, 16, 0.05


def main():
    import ecoz2rs_amd as e
    from tests import oracle_lib

    oracle = oracle_lib.load()
    frames = e.synth.synth_frames(SEED, CLASSES, P, 0, T)
    tmp = tempfile.mkdtemp()
    rc, levels, cbs = oracle.learn(frames, EPS, MAX_M, class_name="_", out_root=tmp)
    assert rc == 0
    for lv in levels:
        name = f"eps_{EPS:g}_M_{lv['M']:04d}.cbook"
        shutil.copy(os.path.join(tmp, "data", "codebooks", "_", name), os.path.join(HERE, "config1_" + name))
    cq = oracle.reflections_to_cq(levels[-1]["reflections"])
    sym, dmin = oracle.quantize(cq, frames)
    seq = os.path.join(HERE, f"config1_M{MAX_M:04d}.seq")
    oracle.L.e2o_seq_save(seq.encode(), b"_", MAX_M, sym.ctypes.data, T)
    meta = {
        "seed": SEED, "classes": CLASSES, "P": P, "T": T, "max_M": MAX_M, "eps": EPS,
        "frames_sha256": hashlib.sha256(frames.tobytes()).hexdigest(),
        "levels": [dict(M=lv["M"], passes=lv["passes"], DD=lv["DD"].hex(), avg=lv["avg"].hex(),
                        sigma=lv["sigma"].hex(), inertia=lv["inertia"].hex(), empty=int(lv["empty"]))
                   for lv in levels],
        "dmin_sum_hex": float(dmin.sum()).hex(),
    }
    json.dump(meta, open(os.path.join(HERE, "config1.json"), "w"), indent=1)
    shutil.rmtree(tmp)
    print("golden fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
