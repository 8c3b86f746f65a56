import os, sys, tempfile, time, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ecoz2rs_amd as e
P, M, NF, TF = 36, 1024, 8, 1_250_000
root = tempfile.mkdtemp(prefix="e2scale_")
os.environ.update(ECOZ2_VQ_OUT_ROOT=root, ECOZ2_VQ_MAX_CODEBOOK_SIZE=str(M), ECOZ2_VQ_QUIET="1", ECOZ2_VQ_TIMING="1")
files = []
for i in range(NF):
    f = os.path.join(root, "data", "predictors", "_", f"{i:05d}.prd")
    e.formats.write_prd(f, "_", e.synth.synth_frames(20243, 20, P, i * TF, TF)); files.append(f)
for rep in range(2):
    t0 = time.time(); e.vq_learn(None, P, 0.05, "_", files); print(f"total {1e3*(time.time()-t0):.1f} ms", flush=True)
shutil.rmtree(root)
