"""Multi-rank runs of the FULL hot path on the GPU: two processes share GPU 0 and talk over gloo
(the collectives are the ones RCCL performs on a multi-GPU node; sharing GPU 0 with the gloo backend is
the development switches of gaussianavatar_amd/parallel.py). Each rank runs AvatarModel.train_stage1 on its
own frames — LBS -> feature net -> skinning -> rasterizer -> L1/DSSIM -> backward -> Adam — and the result
must equal ONE process training on the global batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES = [0, 1, 2, 3]          # the global batch; rank r takes FRAMES[2r : 2r+2]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(batch_size, stage=1):
    from gaussianavatar_amd.avatar_model import AvatarModel, default_params
    # stage 1: 32 -> 64 up-sampling (the fused path texel sharding builds on); stage 2: 64 -> 64, so that the
    # pose encoder's bottleneck still has a few texels per BatchNorm channel
    mp_, npar, op = default_params(batch_size=batch_size, num_points=3000, query_posmap_size=64,
                                   inp_posmap_size=32 if stage == 1 else 64,
                                   image_width=96, image_height=96, num_frames=4, train_stage=stage)
    m = AvatarModel(mp_, npar, op, train=True)
    m.training_setup()
    if stage == 2:
        with torch.no_grad():                       # stand-in for the stage-1 checkpoint: ~1 cm Gaussians
            m.net.decoder.conv8N.bias.fill_(-4.0)
    return m, op


def _trainables(m, stage):
    ps = list(m.net.parameters())
    return ps + (list(m.pose_encoder.parameters()) if stage == 2 else [m.geo_feature])


def _iterate(m, op, frames, steps=2, stage=1):
    """`steps` iterations of the reference's loop body (train.py:66-89) on fixed frames. The gradients are
    taken after step(): that is where the modes that reduce parameter gradients over ranks do it."""
    from gaussianavatar_amd.avatar_model import collate_frames
    from gaussianavatar_amd.losses import l1_loss_w, ssim
    batch = collate_frames([m.train_dataset[i] for i in frames], "cuda")
    gt = torch.ones(len(frames), 3, 96, 96, device="cuda")
    gt[:, :, 30:70, 40:56] = 0.3
    out = {}
    for it in range(steps):
        if stage == 1:
            image, pts, offset_loss, geo_loss, scale_loss = m.train_stage1(batch, 300 + it)
            loss = (op.lambda_scale * scale_loss + op.lambda_rgl * offset_loss + 0.8 * l1_loss_w(image, gt)
                    + 0.2 * (1.0 - ssim(image, gt)) + geo_loss)
        else:
            image, pts, pose_loss, offset_loss = m.train_stage2(batch, 300 + it)
            loss = op.lambda_rgl * offset_loss + 0.8 * l1_loss_w(image, gt) + 0.2 * (1.0 - ssim(image, gt)) + 10 * pose_loss
        m.zero_grad(1)
        loss.backward()
        m.step(1)
        if it == 0:
            out["grads"] = [p.grad.detach().cpu().clone() for p in _trainables(m, stage)]
            out["image"] = image.detach().cpu().clone()
            out["losses"] = [float(offset_loss.detach())]
    out["params"] = [p.detach().cpu().clone() for p in _trainables(m, stage)]
    return out


def _worker(rank, world, port, ret, mode, stage=1, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from gaussianavatar_amd import parallel
    if mode:
        parallel.set_mode(mode)
    # gloo: both ranks share GPU 0 and gloo carries the collectives; nccl (tests/test_parallel_nccl_gpu.py): one GPU
    # per rank over RCCL, the production setup
    parallel.init_from_env(backend=backend)
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    torch.manual_seed(1000 + rank)                 # replicas start DIFFERENT: sync_replicas must fix that
    m, op = _build(2, stage)
    res = _iterate(m, op, FRAMES[2 * rank:2 * rank + 2], stage=stage)
    ret[rank] = res
    parallel.barrier()
    torch.distributed.destroy_process_group()


def _compare(ret, stage):
    r0, r1 = ret[0], ret[1]
    # replicas: identical gradients of the shared parameters and identical parameters after two Adam steps
    for a, b in zip(r0["grads"], r1["grads"]):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max()))
    for a, b in zip(r0["params"], r1["params"]):
        assert float((a - b).abs().max()) <= 1e-6
    # one process, global batch of 4 frames, started from rank 0's initial state (seed 1000)
    torch.manual_seed(1000)
    m, op = _build(4, stage)
    ref = _iterate(m, op, FRAMES, stage=stage)
    img = torch.cat([r0["image"], r1["image"]])
    assert float((img - ref["image"]).abs().mean()) <= 1e-5
    assert abs(r0["losses"][0] - ref["losses"][0]) <= 1e-5 * max(1.0, abs(ref["losses"][0]))
    gmax = max(float(g.abs().max()) for g in ref["grads"])
    for i, (a, b) in enumerate(zip(r0["grads"], ref["grads"])):        # float32 atomics / summation order only
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 2e-4 * gmax, i
    # (parameters after the Adam steps are compared between replicas only: a conv bias in front of a
    # BatchNorm has zero true gradient, Adam turns its rounding noise into full-size steps)


@pytest.mark.parametrize("mode", ["frames", "texels"])
def test_two_ranks_equal_one_process_on_the_global_batch(mode):
    """Stage 1. `frames`: every rank evaluates the batch-invariant decoder, one all-reduce of the output
    gradients. `texels`: the decoder itself is sharded by UV texels (BatchNorm statistics all-reduced, outputs
    assembled with one all-reduce, parameter gradients summed)."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, mode), nprocs=world, join=True)
    _compare(ret, 1)


def test_two_ranks_stage2_synchronised_batchnorm():
    """Stage 2: frames (and with them the decoder's rows and the pose encoder's batch) are spread over the
    ranks; BatchNorm statistics are synchronised, so the result is the single-process batch's."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, None, 2), nprocs=world, join=True)
    _compare(ret, 2)
