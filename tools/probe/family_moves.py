"""After a split, how many frames leave their family?  old = cell at the end of level M/2, new = cell after pass 0 of
level M: in-family even (new == 2 old), in-family odd (new == 2 old + 1), out of family."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ecoz2rs_amd as e
P, S = 36, 1 << 21
os.environ["ECOZ2_VQ_QUIET"] = "1"
frames = e.synth.synth_frames(20244, 20, P, 0, S)
sym = torch.empty(S, dtype=torch.int16, device="cuda")
with e.VqSession(P) as s:
    s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(0.05, 64)
    M = 64
    while M < 1024:
        s.run_pass(sym); s.synchronize()     # assignment under the converged codebook of size M (what the level ended on)
        old = sym.cpu().numpy().astype(np.int64) & 0xffff
        s.grow(); M *= 2
        s.run_pass(sym); s.synchronize()
        new = sym.cpu().numpy().astype(np.int64) & 0xffff
        even = np.mean(new == 2 * old); odd = np.mean(new == 2 * old + 1)
        print(f"M={M}: in-family even {even:.3f}, odd {odd:.3f}, out of family {1 - even - odd:.3f}", flush=True)
        st = s.pass_stats(); s.update()
        s.learn(0.05, M) if False else None
        # finish the level with the real rule
        prev = st.DD
        for _ in range(20):
            s.run_pass(); st = s.pass_stats()
            if not (prev - st.DD) / st.DD >= 0.05: break
            prev = st.DD; s.update()
