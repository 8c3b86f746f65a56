#!/usr/bin/env python3
"""config 3's kernel alone, for rocprofv3: REPS device-resident quantize calls of T frames against a real M = 1024 codebook
(the ladder's, trained on 2^20 of the frames), timed with HIP events on the session's stream.  Prints one JSON line.
   rocprofv3 --kernel-trace --stats -d ... -- python3 tools/quantize_profile.py [T] [REPS]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ecoz2rs_amd as e  # noqa: E402
from ecoz2rs_amd import parallel  # noqa: E402

P, M = 36, 1024
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
os.environ["ECOZ2_VQ_QUIET"] = "1"
frames = e.synth.synth_frames(20243, 20, P, 0, T)
s = e.VqSession(P)
parallel.bind_torch_stream(s, 0)
s.set_frames(frames[: 1 << 20])
s.prepare()
s.init_codebook()
s.learn(0.05, M)
d = torch.from_numpy(frames).cuda()
del frames
sym = torch.empty(T, dtype=torch.int16, device="cuda")
dmin = torch.empty(T, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream(0)
s.quantize_device(d, T, sym, dmin)  # (warm: builds the codebook's limb image)
s.synchronize()
ms = []
for _ in range(REPS):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    s.quantize_device(d, T, sym, dmin)
    b.record(st)
    s.synchronize()
    ms.append(a.elapsed_time(b))
avg = sum(ms) / len(ms)
print(json.dumps({"what": "device-resident quantize (e2vq_quantize_device), config 3 size", "frames": T, "codebook_size": M, "reps": REPS,
                  "event_ms": ms, "avg_ms": avg, "frames_per_sec": T / (avg * 1e-3),
                  "f16_mfma_tflops_executed": 2 * 16 * 15.0 * M * T / (avg * 1e-3) / 1e12,
                  "algorithmic_bytes_per_launch": 298 * T, "kernel_sources_sha16": __import__("bench").kernel_sources_sha16()}), flush=True)
