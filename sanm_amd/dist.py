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


class TestXfer(ctypes.Structure):
    """sanm_test_xfer (include/sanm_hip_test.h)"""
    _fields_ = [("src", ctypes.c_int32), ("dst", ctypes.c_int32), ("off", ctypes.c_int64), ("cnt", ctypes.c_int64),
                ("src_stage", ctypes.c_int32)]


_P2P_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(TestXfer), ctypes.c_int)


def set_staged_p2p(api, log=None):
    """set_host_p2p for DEVICE memory: every transfer through a host copy and a gloo group -- several ranks on ONE GPU run
    the point-to-point branch of the distributed solver on the HIP backend (RCCL refuses two ranks on one device)."""
    return set_host_p2p(api, log, staged=True)


def set_host_p2p(api, log=None, staged=False):
    """the distributed direct solver's point-to-point branch over torch.distributed on HOST memory (gloo; the host harness):
    sanm_test_set_p2p with a callback that posts every transfer of the list this rank takes part in (isend / irecv,
    broadcasts from their source) and waits for them.  `log`, a list, receives (sends, receives, broadcasts, doubles)
    per call.  Returns the callback object: keep it alive while solvers use it."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()

    def view(base, off, cnt):
        addr = ctypes.addressof(base.contents) + 8 * int(off)
        if staged:
            return torch.as_tensor(_DevicePtr(addr, cnt), device="cuda")
        return torch.from_numpy(np.frombuffer((ctypes.c_double * int(cnt)).from_address(addr), dtype=np.float64))

    def p2p(_user, base, xfers, n):
        try:
            reqs, stat, back = [], [0, 0, 0, 0], []
            for i in range(n):
                x = xfers[i]
                if x.cnt == 0:
                    continue
                takes_part = x.dst < 0 or (x.src != x.dst and rank in (x.src, x.dst))
                if not takes_part:
                    continue
                dev = view(base, x.off, x.cnt)
                buf = dev.cpu() if staged else dev  # (staged: the host copy travels, receivers copy it back below)
                if x.dst < 0:
                    reqs.append(dist.broadcast(buf, src=x.src, async_op=True))
                    stat[2] += 1
                    receives = rank != x.src
                elif rank == x.src:
                    reqs.append(dist.isend(buf, dst=x.dst, tag=i))
                    stat[0] += 1
                    receives = False
                else:
                    reqs.append(dist.irecv(buf, src=x.src, tag=i))
                    stat[1] += 1
                    receives = True
                if staged and receives:
                    back.append((dev, buf))
                stat[3] += int(x.cnt)
            for r in reqs:
                r.wait()
            for dev, buf in back:
                dev.copy_(buf)
            if staged:
                torch.cuda.current_stream().synchronize()
            if log is not None:
                log.append(tuple(stat))
            return 0
        except Exception:  # (never an exception through the C frames)
            import traceback
            traceback.print_exc()
            return 1

    cb = _P2P_FN(p2p)
    api.lib.sanm_test_set_p2p.restype = ctypes.c_int
    api.lib.sanm_test_set_p2p.argtypes = [_P2P_FN, ctypes.c_void_p]
    api.check(api.lib.sanm_test_set_p2p(cb, None))
    return cb

