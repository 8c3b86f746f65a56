"""Frame sharding + the per-iteration exchange for N ranks (one process per GPU).

Frames shard contiguously by global index; every LBG iteration all-reduces the per-cell exact
integer sums (int64) -- RCCL over xGMI when the process group is `nccl`, staged through host
memory for `gloo` (tests).  Integer sums are associative, so the reduced rows -- and every
codebook derived from them -- are bit-identical for any rank count.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` frames for `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def bind_torch_stream(session, device_index):
    """Run the session on a dedicated torch stream and make it the thread's current stream, so torch ops
    (the collective, host copies) and the session's kernels are ordered on ONE stream.  (torch's default
    stream has handle 0, which the C-ABI reads as "use the session's own stream" -- hence a real stream.)"""
    st = torch.cuda.Stream(device=device_index)
    torch.cuda.set_stream(st)
    session.set_stream(st.cuda_stream)
    return st


class _DeviceBuffer:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def reduce_int64_(t, op, group=None):
    """In-place all-reduce of an int64 tensor: op 0 = sum, 1 = max (non-negative bit patterns)."""
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=group)
    return t


def make_allreduce(device_index, group=None):
    """Hook for VqSession.set_allreduce: reduces `count` int64 at device address `ptr` in place."""
    backend = dist.get_backend(group)
    views = {}  # (ptr, count) -> tensor view: the session reduces the same few buffers every iteration

    def hook(ptr, count, op, _stream):
        t = views.get((ptr, count))
        if t is None:
            if len(views) > 64:
                views.clear()
            t = views[(ptr, count)] = torch.as_tensor(_DeviceBuffer(ptr, count), device=f"cuda:{device_index}")
        if backend == "nccl":
            reduce_int64_(t, op, group)  # RCCL, ordered after the session's work on the current stream
        else:
            h = t.cpu()
            reduce_int64_(h, op, group)
            t.copy_(h)

    return hook
