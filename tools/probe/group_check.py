import os, sys, time, tempfile, filecmp
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
P, T = 36, 1 << 20
frames = e.synth.synth_frames(20245, 20, P, 0, T)
os.environ["ECOZ2_VQ_QUIET"] = "1"; os.environ["ECOZ2_VQ_MAX_CODEBOOK_SIZE"] = "1024"
outs = {}
for ranks in (1, 4, 8):
    root = tempfile.mkdtemp()
    f = os.path.join(root, "all.prd"); e.formats.write_prd(f, "_", frames)
    os.environ["ECOZ2_VQ_OUT_ROOT"] = root; os.environ["ECOZ2_VQ_GPUS"] = str(ranks)
    t0 = time.time(); e.vq_learn(None, P, 0.05, "_", [f]); dt = time.time() - t0
    outs[ranks] = root
    print(f"ranks={ranks}: ecoz2_vq_learn 2..1024 on {T} frames: {dt:.2f}s wall (file read + upload + learn)", flush=True)
for ranks in (4, 8):
    for M in (2, 64, 256, 1024):
        a = os.path.join(outs[1], "data/codebooks/_", f"eps_0.05_M_{M:04d}.cbook"); b = a.replace(outs[1], outs[ranks])
        assert filecmp.cmp(a, b, shallow=False), (ranks, M)
print("codebooks of 4 and 8 in-process ranks are byte-identical to the single-rank ones")
