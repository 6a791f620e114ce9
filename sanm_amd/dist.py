"""All-reduce callbacks for the tet-sharded solver (one process per GPU).

``torch.distributed`` is the communication layer: backend "nccl" is RCCL on ROCm
and runs over xGMI inside a node.  The C ABI hands the callback a raw device
pointer; it is wrapped as a tensor without a copy.  The solver synchronises its
own stream before calling; the callback returns only when the reduction is
complete.
"""
from __future__ import annotations

import ctypes


def init_native_comm(api, rank, world, device="cuda"):
    """The library's own RCCL communicator (sanm_hip_comm_init): ncclAllReduce is queued on the solver's stream,
    so the sharded order loop runs without host synchronisation.  The 128-byte identifier travels from rank 0
    to the others through torch.distributed's default group (any out-of-band channel would do).

    Returns True when EVERY rank holds the communicator, False when all ranks agreed to use the callback path
    instead.  Every rank takes the same sequence of torch.distributed collectives whatever fails where:

      1. each rank probes whether RCCL can be loaded from C++ (no collective involved); rank 0 also draws the
         identifier -- a failure there becomes "not available", not an exception that skips the broadcast;
      2. broadcast of the identifier (None on failure), MIN all-reduce of the availability flags;
      3. only if all ranks are able: ncclCommInitRank (itself a collective), then a MIN all-reduce of its
         outcomes; a rank whose init failed makes every rank drop its communicator again.
    """
    ok = 1 if api.comm_available() else 0
    uid = [None]
    if rank == 0 and ok:
        try:
            uid[0] = api.comm_unique_id()
        except Exception:  # noqa: BLE001 -- reported through the flag below
            ok = 0
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok or uid[0] is None:
        return False
    try:
        api.comm_init(rank, world, uid[0])
    except Exception:  # noqa: BLE001
        ok = 0
    if dist is not None:
        import torch
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if ok and not int(flag.item()):
            api.comm_destroy()
        ok = int(flag.item())
    return bool(ok)


class _DevicePtr:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


def make_rccl_allreduce():
    """sum over ranks with the default process group (backend nccl = RCCL)."""
    import torch
    import torch.distributed as dist

    def allreduce(ptr, count):
        t = torch.as_tensor(_DevicePtr(ptr, count), device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        torch.cuda.current_stream().synchronize()

    return allreduce


def make_host_allreduce():
    """the same on host memory (gloo); used by the CPU tests of the sharded path."""
    import numpy as np
    import torch
    import torch.distributed as dist

    def allreduce(ptr, count):
        buf = (ctypes.c_double * int(count)).from_address(int(ptr))
        a = np.frombuffer(buf, dtype=np.float64)
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    return allreduce


def make_staged_allreduce():
    """sum over ranks of a DEVICE buffer through a host-side process group (gloo): device -> pinned host copy,
    all-reduce, copy back.  Not a fast path: it lets several ranks share ONE GPU (RCCL refuses two ranks on one
    device), which is how the world > 1 branch of the sharded HIP path is exercised on a single-GPU test box."""
    import torch
    import torch.distributed as dist

    def allreduce(ptr, count):
        t = torch.as_tensor(_DevicePtr(ptr, count), device="cuda")
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        torch.cuda.current_stream().synchronize()

    return allreduce
