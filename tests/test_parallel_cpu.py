"""N > 1 path on CPU (gloo, world_size 2): frame sharding + integer all-reduce of the cell sums.
The compute stand-in here is the oracle (tests only); the product's GPU ranks run the same exchange."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ecoz2rs_amd import parallel


def test_shard_range_partitions_exactly():
    for total, world in ((10, 3), (16 << 20, 8), (5, 8), (1, 1)):
        cuts = [parallel.shard_range(total, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ecoz2rs_amd as e
    from tests import oracle_lib

    oracle = oracle_lib.load()
    P, T, M = 36, 3001, 8
    lo, hi = parallel.shard_range(T, rank, world)
    shard = e.synth.synth_frames(77, 3, P, lo, hi - lo)  # each rank generates only its own frames
    # data statistics: max via MAX-reduce of the bit pattern, sums via SUM-reduce (as e2vq_prepare does)
    mx = torch.tensor([np.abs(shard).max()], dtype=torch.float64).view(torch.int64)
    parallel.reduce_int64_(mx, 1)
    maxabs = float(mx.view(torch.float64)[0])
    sh_r, _ = oracle.shifts(maxabs)
    refl = np.zeros((M, P + 1))
    refl[:, 1:] = np.linspace(-0.3, 0.3, M)[:, None] * 0.9 ** np.arange(P)[None, :]
    cq = oracle.reflections_to_cq(refl)
    Ed = oracle.dist_exponent(cq, maxabs)
    _s, _d, rows = oracle.run_pass(cq, shard, sh_r, Ed)
    t = torch.from_numpy(rows.reshape(-1).copy())
    parallel.reduce_int64_(t, 0)
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), t.numpy().reshape(rows.shape))
    np.save(os.path.join(out_dir, f"maxabs_{rank}.npy"), np.array([maxabs]))
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_rank(tmp_path, oracle):
    import ecoz2rs_amd as e

    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    P, T, M = 36, 3001, 8
    frames = e.synth.synth_frames(77, 3, P, 0, T)
    maxabs = float(np.abs(frames).max())
    sh_r, _ = oracle.shifts(maxabs)
    refl = np.zeros((M, P + 1))
    refl[:, 1:] = np.linspace(-0.3, 0.3, M)[:, None] * 0.9 ** np.arange(P)[None, :]
    cq = oracle.reflections_to_cq(refl)
    _s, _d, rows = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, maxabs))
    for r in range(world):
        assert float(np.load(tmp_path / f"maxabs_{r}.npy")[0]) == maxabs
        assert np.array_equal(np.load(tmp_path / f"rows_{r}.npy"), rows)  # bit-identical for any rank count
