import os, sys, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import ecoz2rs_amd as e
P, M, S = 36, 1024, 1 << 21
frames = e.synth.synth_frames(20244, 20, P, 0, S)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(0.05, 512); s.grow()
s.enable_timing(True)
ts = []
for i in range(12):
    s.run_pass(None, None); ts.append(s.last_pass_kernel_ms()); s.pass_stats(); s.update()
print("stagger", os.environ.get("ECOZ2_VQ_STAGGER", "0"), "kernel ms", np.round(ts[2:], 3), "mean", round(float(np.mean(ts[2:])), 4))
'''
variants = sys.argv[1:] or ["default"]
for rnd in range(2):
    for v in variants:
        env = dict(os.environ)
        if v.startswith("stagger="):
            env["ECOZ2_VQ_STAGGER"] = v.split("=")[1]
        elif v != "default":
            env["ECOZ2VQ_LIB"] = os.path.join(root, "tools", "probe", "ab", v)
        print("variant:", v, flush=True)
        subprocess.run([sys.executable, "-c", code, root], env=env)
