"""The multi-rank runs of tests/test_parallel_gpu.py over the PRODUCTION transport: one process per GPU,
`torch.distributed` backend "nccl" (= RCCL on ROCm) — frames mode, texels mode and stage 2 each equal to ONE process
on the global batch. Needs two HIP devices: skipped on the 1-GPU boxes the round's GPU tests run on, so that the first
RCCL process group this code creates is not the one inside a benchmark."""
import pytest
import torch
import torch.multiprocessing as mp

from tests.test_parallel_gpu import _compare, _free_port, _worker

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices (RCCL over xGMI)")]


@pytest.mark.parametrize("mode,stage", [("frames", 1), ("texels", 1), (None, 2)])
def test_two_gpus_over_rccl_equal_one_process(mode, stage):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, mode, stage, "nccl"), nprocs=world, join=True)
    _compare(ret, stage)
