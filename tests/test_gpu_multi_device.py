"""RCCL with more than one rank on more than one device (BASELINE config 4).  Every test here is skipped unless the
box shows at least two HIP devices: the 1-GPU boxes of `pytest -m gpu` rehearse the same code with ranks sharing a device
(tests/test_gpu_cabi.py), and only these put a real wire under the int64 all-reduce."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import ecoz2rs_amd as e

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 36


def _ndev():
    try:
        return e.lib.e2vq_device_count()
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_ndev() < 2, reason="needs two HIP devices (RCCL wants one device per rank)")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import ecoz2rs_amd as e
from ecoz2rs_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = rank % torch.cuda.device_count()
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev}"))
T, MAXM, SEED = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
lo, hi = parallel.shard_range(T, rank, world)
frames = e.synth.synth_frames(SEED, 5, 36, lo, hi - lo)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(36, device=dev)
parallel.bind_torch_stream(s, dev)
s.set_allreduce(parallel.make_allreduce(dev), rank, world)
s.set_frames(frames); s.prepare(); s.init_codebook()
levels = s.learn(0.05, MAXM)
np.save(sys.argv[2] + f"/cb_{rank}.npy", s.get_codebook())
np.save(sys.argv[2] + f"/passes_{rank}.npy", np.array([l.passes for l in levels]))
np.save(sys.argv[2] + f"/dev_{rank}.npy", np.array([dev]))
s.close(); dist.barrier(); dist.destroy_process_group()
"""


def _single_rank(frames, max_m):
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, max_m)
        return levels, s.get_codebook()


@needs_two
@pytest.mark.parametrize("world", [2, 4])
def test_process_per_gpu_over_rccl_equals_single_rank_and_oracle(tmp_path, oracle, world):
    """One process per GPU, torch.distributed `nccl` = RCCL over xGMI: the all-reduce hook bench.py runs, at config 4's
    codebook size.  Codebook of every rank == single rank == oracle, bit for bit; pass counts and DD too."""
    if _ndev() < world:
        pytest.skip(f"{_ndev()} device(s) < {world} ranks")
    if world > 6:
        pytest.skip("the GPU box allows six of our processes on its cards at once")
    T, max_m, seed = 200_003, 1024, 61
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path), str(T), str(max_m), str(seed)], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    assert len({int(np.load(tmp_path / f"dev_{r}.npy")[0]) for r in range(world)}) == world  # a device of its own for every rank
    frames = e.synth.synth_frames(seed, 5, P, 0, T)
    levels, ref = _single_rank(frames, max_m)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"cb_{r}.npy").view(np.uint64), ref.view(np.uint64))
        assert list(np.load(tmp_path / f"passes_{r}.npy")) == [l.passes for l in levels]
    rc, levels_o, _cbs = oracle.learn(frames, 0.05, max_m)
    assert rc == 0 and [l.passes for l in levels] == [lv["passes"] for lv in levels_o]
    assert np.array_equal(ref.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
    assert [l.DD for l in levels] == [lv["DD"] for lv in levels_o]


@needs_two
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_in_library_rccl_group_equals_single_rank_and_oracle(tmp_path, oracle, ranks):
    """ECOZ2_VQ_GPUS=N behind the reference's single-process entry point: the library's own group, RCCL loaded with dlopen,
    ncclCommInitAll over N devices, ncclAllReduce(int64) per pass -- M = 1024, codebook bytes == single rank == oracle.
    (A process of its own, as under the Rust host: this test process has PyTorch's bundled ROCm libraries mapped.)"""
    if _ndev() < ranks:
        pytest.skip(f"{_ndev()} device(s) < {ranks} ranks")
    T, max_m, seed = 200_003, 1024, 62
    frames = e.synth.synth_frames(seed, 5, P, 0, T)
    f = tmp_path / "data" / "predictors" / "_" / "all.prd"
    e.formats.write_prd(str(f), "_", frames)
    env = dict(os.environ, ECOZ2_VQ_OUT_ROOT=str(tmp_path), ECOZ2_VQ_MAX_CODEBOOK_SIZE=str(max_m), ECOZ2_VQ_GPUS=str(ranks),
               ECOZ2_VQ_COLLECTIVE="rccl", NCCL_DEBUG="WARN")
    env.pop("ECOZ2_VQ_QUIET", None)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import ecoz2rs_amd as e; seen = []; "
            f"e.vq_learn(None, {P}, 0.05, '_', [{str(f)!r}], callback=lambda *a: seen.append(a)); "
            "print('SEEN', [(s[0], s[1].hex(), s[2].hex(), s[3].hex()) for s in seen])")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "collective: RCCL" in r.stdout and "ncclAllReduce(int64 sum)" in r.stdout
    assert "peer-to-peer" not in r.stdout.split("collective:")[1].splitlines()[0]
    seen = eval([ln for ln in r.stdout.splitlines() if ln.startswith("SEEN ")][0][5:])
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, max_m)
    assert rc == 0
    assert seen == [(c[0], float(c[1]).hex(), float(c[2]).hex(), float(c[3]).hex()) for c in cbs_o]
    _c, _p, refl = e.formats.read_cbook(str(tmp_path / "data" / "codebooks" / "_" / f"eps_0.05_M_{max_m:04d}.cbook"))
    assert np.array_equal(refl.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
    _levels, ref = _single_rank(frames, max_m)
    assert np.array_equal(refl.view(np.uint64), ref.view(np.uint64))


@needs_two
def test_bench_two_gpus_reports_distinct_devices_and_per_level_exchange():
    """`python bench.py --gpus 2` as the driver's SCALE run starts it (self-launched here): two devices, nccl, strong scaling
    of config 4 (2^23 frames per GPU), the per-level table with the exchange's device time per pass."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["frames_per_gpu"] == (1 << 24) // 2
    c = d["config"]["collective"]
    assert c["distinct_devices"] == 2 and c["backend"].startswith("nccl") and c["allreduce_us_per_call"] > 0
    assert d["config"]["parity"]["ok"]
    lv = d["config"]["learn_end_to_end"]["levels"]
    assert all("allreduce_ms_per_step" in x and x["allreduce_ms_per_step"] < x["step_ms"] for x in lv)
