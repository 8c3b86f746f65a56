"""BASELINE-sized runs through the reference's own entry points (files on disk, one process):
config 3: ecoz2_vq_quantize of 10 M frames (8 .prd files) against an M = 1024 codebook, ECOZ2_VQ_GPUS = 1 and 4 workers
config 4 (one GPU's worth): ecoz2_vq_learn on the same files up to M = 1024 with ECOZ2_VQ_GPUS = 1 and 4 in-process ranks
(ranks / workers beyond the device count share the GPU).  Checks: identical .seq / .cbook bytes for any N; prints timings."""
import hashlib, os, shutil, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e

P, M, NF, TF = 36, 1024, 8, 1_250_000
root = tempfile.mkdtemp(prefix="e2scale_")
os.environ["ECOZ2_VQ_OUT_ROOT"] = root
os.environ["ECOZ2_VQ_MAX_CODEBOOK_SIZE"] = str(M)
os.environ["ECOZ2_VQ_QUIET"] = "1"
t0 = time.time()
files = []
for i in range(NF):
    f = os.path.join(root, "data", "predictors", "_", f"{i:05d}.prd")
    e.formats.write_prd(f, "_", e.synth.synth_frames(20243, 20, P, i * TF, TF))
    files.append(f)
print(f"wrote {NF} files x {TF} frames ({NF*TF*296/1e9:.2f} GB) in {time.time()-t0:.1f} s", flush=True)


def digest(pattern_dir, ext):
    h = hashlib.sha256()
    for dp, _dn, fn in sorted(os.walk(pattern_dir)):
        for name in sorted(fn):
            if name.endswith(ext):
                h.update(name.encode()); h.update(open(os.path.join(dp, name), "rb").read())
    return h.hexdigest()[:16]


cb_digest = {}
for gpus in (1, 4):
    os.environ["ECOZ2_VQ_GPUS"] = str(gpus)
    shutil.rmtree(os.path.join(root, "data", "codebooks"), ignore_errors=True)
    t0 = time.time()
    e.vq_learn(None, P, 0.05, "_", files)
    dt = time.time() - t0
    cb_digest[gpus] = digest(os.path.join(root, "data", "codebooks"), ".cbook")
    print(f"vq learn 2..{M} on {NF*TF} frames, {gpus} in-process rank(s): {dt:.2f} s wall incl. file reads + upload "
          f"= {NF*TF/dt/1e6:.1f} M frames/s; codebooks sha {cb_digest[gpus]}", flush=True)
assert cb_digest[1] == cb_digest[4], "codebooks differ between 1 and 4 ranks"
cbook = os.path.join(root, "data", "codebooks", "_", f"eps_0.05_M_{M:04d}.cbook")
seq_digest = {}
for gpus in (1, 4):
    os.environ["ECOZ2_VQ_GPUS"] = str(gpus)
    shutil.rmtree(os.path.join(root, "data", "sequences"), ignore_errors=True)
    t0 = time.time()
    e.vq_quantize(cbook, files, False)
    dt = time.time() - t0
    seq_digest[gpus] = digest(os.path.join(root, "data", "sequences"), ".seq")
    print(f"vq quantize {NF*TF} frames vs M={M}, {gpus} worker(s): {dt:.2f} s wall incl. file reads, H2D, .seq writes "
          f"= {NF*TF/dt/1e6:.1f} M frames/s; .seq sha {seq_digest[gpus]}", flush=True)
assert seq_digest[1] == seq_digest[4], ".seq files differ between 1 and 4 workers"
# a corpus of many short recordings (notes.md: thousands of files of a few thousand frames): 5 000 files x 2 000 frames
shutil.rmtree(os.path.join(root, "data", "predictors"), ignore_errors=True)
NS, TS = 5000, 2000
t0 = time.time()
small = []
block = e.synth.synth_frames(20243, 20, P, 0, NS * TS)
for i in range(NS):
    f = os.path.join(root, "data", "predictors", f"c{i % 20}", f"{i:05d}.prd")
    e.formats.write_prd(f, f"c{i % 20}", block[i * TS:(i + 1) * TS])
    small.append(f)
del block
print(f"wrote {NS} files x {TS} frames in {time.time()-t0:.1f} s", flush=True)
seq_small = {}
for gpus in (1, 4):
    os.environ["ECOZ2_VQ_GPUS"] = str(gpus)
    shutil.rmtree(os.path.join(root, "data", "sequences"), ignore_errors=True)
    t0 = time.time()
    e.vq_quantize(cbook, small, False)
    dt = time.time() - t0
    seq_small[gpus] = digest(os.path.join(root, "data", "sequences"), ".seq")
    print(f"vq quantize {NS} files x {TS} frames vs M={M}, {gpus} worker(s): {dt:.2f} s wall = {NS*TS/dt/1e6:.1f} M frames/s, "
          f"{NS/dt:.0f} files/s; .seq sha {seq_small[gpus]}", flush=True)
assert seq_small[1] == seq_small[4]
shutil.rmtree(root)
print("scale check ok: identical bytes for 1 and 4 ranks / workers")
