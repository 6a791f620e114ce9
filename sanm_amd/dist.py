"""All-reduce callbacks for the tet-sharded solver (one process per GPU).

``torch.distributed`` is the communication layer: backend "nccl" is RCCL on ROCm
and runs over xGMI inside a node.  The C ABI hands the callback a raw device
pointer; it is wrapped as a tensor without a copy.  The solver synchronises its
own stream before calling; the callback returns only when the reduction is
complete.
"""
from __future__ import annotations

import ctypes


def init_native_comm(api, rank, world):
    """The library's own RCCL communicator (sanm_hip_comm_init): ncclAllReduce is queued on the solver's stream,
    so the sharded order loop runs without host synchronisation.  The 128-byte identifier travels from rank 0
    to the others through torch.distributed's default group (any out-of-band channel would do)."""
    uid = [api.comm_unique_id() if rank == 0 else None]
    if world > 1:
        import torch.distributed as dist
        dist.broadcast_object_list(uid, src=0)
    api.comm_init(rank, world, uid[0])


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
