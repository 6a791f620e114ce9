"""The RCCL all-reduce callback of the tet-sharded path on a real GPU: a
single-rank process group (only one GPU is available to the test box) checks the
zero-copy device-pointer wrapping and the collective call itself."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_allreduce_callback_single_rank():
    import torch
    import torch.distributed as dist
    from sanm_amd import dist as sdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        fn = sdist.make_rccl_allreduce()
        x = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
        fn(x.data_ptr(), x.numel())
        assert np.array_equal(x.cpu().numpy(), np.arange(1000) * 0.5)
    finally:
        dist.destroy_process_group()


def _cfg():
    # (Pade off: the step count of two runs whose nodal sums are formed in different orders is only guaranteed equal
    # without its ill-conditioned decisions -- tests/lockstep.py; tests/test_sharded.py runs both settings)
    return {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0}, "g": [0, -9.81, 0],
            "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 12,
            "disable_pade": True}


def test_sharded_hip_path_with_the_library_communicator_equals_the_unsharded_solve():
    """sanm_anm_eqn_solver_create_sharded on the HIP backend with the library's own RCCL communicator
    (ncclAllReduce queued on the solver's stream): one rank, so every collective is the identity and the result
    must equal the unsharded solve -- same step count, same vertices.  The test box has one GPU; what this proves
    is that the sharded code path (tet range, per-order all-reduce of b_k, all-reduce of the Jacobian values and
    of f(x0), RCCL bound by dlopen, solver stream) runs on the device."""
    import sanm_amd
    from sanm_amd import dist as sdist
    from sanm_amd import fea as dfea
    api = sanm_amd.get_api(0)
    assert api.backend_name() == "hip"
    sdist.init_native_comm(api, 0, 1)
    try:
        mesh = lambda: dfea.make_cuboid(8, 4, 4, 0.025)
        ref = dfea.GravityRun(api, mesh(), _cfg(), solver_rtol=1e-15).run()
        run = dfea.GravityRun(api, mesh(), _cfg(), shard=(0, 1, None), solver_rtol=1e-15).run()
        assert run.solver.get_nr_iter() == ref.solver.get_nr_iter()
        V, Vr = run.vertices(), ref.vertices()
        assert np.abs(V - Vr).max() / np.abs(Vr).max() < 1e-9
        assert run.rms[-1] < 1e-10
    finally:
        api.comm_destroy()


def test_sharded_hip_path_with_the_callback_equals_the_unsharded_solve():
    """the same through the C ABI's all-reduce callback (torch.distributed nccl group of one rank)"""
    import torch
    import torch.distributed as dist
    import sanm_amd
    from sanm_amd import dist as sdist
    from sanm_amd import fea as dfea
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        api = sanm_amd.get_api(0)
        ncall = [0]
        base = sdist.make_rccl_allreduce()

        def counted(ptr, count):
            ncall[0] += 1
            base(ptr, count)

        mesh = lambda: dfea.make_cuboid(8, 4, 4, 0.025)
        ref = dfea.GravityRun(api, mesh(), _cfg(), solver_rtol=1e-15).run()
        run = dfea.GravityRun(api, mesh(), _cfg(), shard=(0, 1, counted), solver_rtol=1e-15).run()
        steps = run.solver.get_nr_iter()
        assert steps == ref.solver.get_nr_iter()
        assert np.abs(run.vertices() - ref.vertices()).max() / np.abs(ref.vertices()).max() < 1e-9
        assert ncall[0] == steps * (1 + 1 + (12 - 1)) + 1
    finally:
        dist.destroy_process_group()
