"""Where does the evaluate phase of the prefiltered pass go?  One converged M = 512 codebook (made by the product library,
cached in /tmp), split to 1024, then timed passes on that fixed codebook: full accumulate (ECOZ2_VQ_INCREMENTAL=0),
incremental with nothing to move, assignment only.  ECOZ2VQ_LIB picks an ablated variant (tools/probe/ab:
-DE2VQ_PRE_ABLATE_VERIFY / _ROUND2 / _BF: wrong results, timing only)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
P, S = 36, 1 << 21
os.environ["ECOZ2_VQ_QUIET"] = "1"
frames = e.synth.synth_frames(20244, 20, P, 0, S)
cache = "/tmp/e_phase_cb512.npy"
if not os.path.exists(cache):
    assert "ECOZ2VQ_LIB" not in os.environ, "make the codebook with the product library first"
    with e.VqSession(P) as s:
        s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(0.05, 512); np.save(cache, s.get_codebook())
tag = os.path.basename(os.path.dirname(os.environ.get("ECOZ2VQ_LIB", "/product/x")))
for name, env, mode in (("full", "0", "2"), ("incr-idle", "1", "2"), ("assign", "1", "0")):
    os.environ["ECOZ2_VQ_INCREMENTAL"] = env  # (read when the session is created)
    os.environ["ECOZ2_VQ_FORCE_MODE"] = mode
    with e.VqSession(P) as s:
        s.set_frames(frames); s.prepare(); s.set_codebook(np.load(cache)); s.grow()
        s.enable_timing(True)
        ts = []
        for i in range(60):  # (the clocks take ~30 passes to settle)
            s.run_pass(); ts.append(s.last_pass_kernel_ms())
        print(f"{tag:12s} {name:10s} kernel ms: median of the last 25 = {np.median(ts[35:]):.3f}  (min {min(ts[35:]):.3f})  fallback {s.last_pass_info()[1]}", flush=True)
