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
