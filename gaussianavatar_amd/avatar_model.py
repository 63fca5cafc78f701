"""`AvatarModel` — the orchestrator of the render-and-fit hot path with the method surface of
/root/reference/model/avatar_model.py (so the reference's train.py / eval.py /
render_novel_pose.py loops drive it unchanged):

    AvatarModel(model_parms, net_parms, opt_parms, load_iteration=None, train=True)
    training_setup() · zero_grad(epoch) · step(epoch)
    train_stage1(batch, it) -> (image[B,3,H,W], full_pred[B,N,3], offset_loss, geo_loss, scale_loss)
    train_stage2(batch, it) -> (image, full_pred, pose_loss, offset_loss)
    render_free_stage1(batch, it) / render_free_stage2(batch, it) -> image[B,3,H,W]
    save(it) · load(it) · stage_load(path) · stage2_load(epoch)      (same dict keys / file names)
    getTrainDataloader() · getTestDataset() · getNovelposeDataset() · getNovelviewDataset()

What is different underneath (all output-preserving):
  * pose -> joint transforms -> `@ inv_mats` is ONE HIP launch (lbs.joint_transforms) and the two
    skinning einsums are one fused HIP kernel (lbs.skin) — no [B,N,4,4] intermediate;
  * the decoder runs once per iteration in stage 1 (its input is batch-invariant) and works
    point-major; the valid-texel gather happens on that single copy;
  * camera scalars stay python numbers (no .item() syncs), `uv_coord_map` carries no grad;
  * data parallelism: frames are sharded over ranks; ONE RCCL all-reduce of the per-Gaussian
    output gradients [N,7] per iteration keeps the replicas bit-identical (parallel.py).
Assets come from disk in the reference's formats when `model_parms.source_path` holds a dataset
(dataset.py: MonoDataset_*, load_assets — SMPL files, uv masks, lbs maps, query posmaps, frames),
and from the seeded in-memory generator (synthetic.make_assets) otherwise, because none of the
reference's assets ship with it.
"""
from __future__ import annotations

import os
import weakref
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _dev, fused, parallel
from .lbs import SMPLBody, skin
from .network import POP_no_unet, UnetNoCond5DS
from .renderer import render_batch, render_frames  # noqa: F401
from .synthetic import make_assets, make_frames
from . import dataset as disk


def default_params(**overrides):
    """(model_parms, net_parms, opt_parms) with the reference's defaults
    (/root/reference/arguments/__init__.py:55-142). Extra synthetic-only knobs: num_points,
    num_frames, image_width, image_height."""
    model = SimpleNamespace(
        source_path="", model_path="./output/synthetic", project_path=os.getcwd(), stage1_out_path="",
        smpl_model_path=os.getcwd() + "/assets/smpl_files/smpl", smplx_model_path=os.getcwd() + "/assets/smpl_files/smplx",
        test_folder=os.getcwd() + "/assets/test_pose",
        save_epoch=30, train_stage=1, dataset_type="synthetic", smpl_gender="neutral", smpl_type="smpl",
        no_mask=0, fixed_inp=0, train_mode=0, cam_static=1, white_background=True,
        bullet_pose_list=[112, 217, 755], batch_size=2, query_posmap_size=512, inp_posmap_size=128,
        num_points=200_000, num_frames=16, image_width=1024, image_height=1024)
    net = SimpleNamespace(c_pose=64, c_geom=64, hsize=128, nf=32, up_mode="upconv", use_dropout=0,
                          pos_encoding=0, num_emb_freqs=6, posemb_incl_input=0, geom_layer_type="conv",
                          gaussian_kernel_size=5)
    epochs = 200
    opt = SimpleNamespace(epochs=epochs, lambda_dssim=0.2, lambda_scale=3e-2, lambda_lpips=0.2,
                          lambda_pose=10, lambda_rgl=1e1, log_iter=2000, lpips_start_iter=30,
                          pose_op_start_iter=1800, lr_net=3e-3, lr_geomfeat=5e-4,
                          sched_milestones=[int(epochs / 3), int(epochs * 2 / 3)])
    for k, v in overrides.items():
        for ns in (model, net, opt):
            if hasattr(ns, k):
                setattr(ns, k, v)
                break
        else:
            raise KeyError(k)
    return model, net, opt


class SyntheticFrames(torch.utils.data.Dataset):
    """Dataset items with the keys of MonoDataset_train.__getitem__
    (/root/reference/scene/dataset_mono.py:224-257). Camera scalars are python numbers."""

    def __init__(self, frames: dict, stage: int, inp_posmap_size: int, test: bool = False):
        self.frames, self.stage, self.test = frames, stage, test
        cam = frames["camera"]
        self.cam = {k: torch.tensor(cam[k]) for k in ("world_view_transform", "full_proj_transform", "camera_center")}
        self.cam.update(FovX=cam["FovX"], FovY=cam["FovY"], width=cam["width"], height=cam["height"])
        g = torch.Generator().manual_seed(1234)
        self.inp = torch.randn(frames["pose"].shape[0], 3, inp_posmap_size, inp_posmap_size, generator=g) * 0.3
        self.gt = None

    def __len__(self):
        return self.frames["pose"].shape[0]

    def __getitem__(self, i):
        item = dict(self.cam)
        item["pose_idx"] = i
        item["original_image"] = self.gt[i] if self.gt is not None else torch.ones(3, self.cam["height"], self.cam["width"])
        if self.stage == 2:
            item["inp_pos_map"] = self.inp[i]
        if self.test:
            item["pose_data"] = self.frames["pose"][i]
            item["transl_data"] = self.frames["transl"][i]
        item["rest_pose"] = self.frames["rest_pose"][i]
        return item


def collate_frames(items, device="cuda"):
    """Collate to the batch dict the model consumes; tensors go to `device` (None: stay on the
    host, for loader worker processes), camera scalars stay python lists (the reference's to_cuda
    turns them into 0-d CUDA tensors -> host syncs)."""
    out = {}
    for k in items[0]:
        v0 = items[0][k]
        if isinstance(v0, np.ndarray):
            out[k] = torch.from_numpy(np.stack([it[k] for it in items])).float()
        elif torch.is_tensor(v0):
            out[k] = torch.stack([it[k] for it in items])
        elif k == "pose_idx":
            out[k] = torch.tensor([it[k] for it in items], dtype=torch.long)
        else:
            out[k] = [it[k] for it in items]
        if device is not None and torch.is_tensor(out[k]):
            out[k] = out[k].to(device, non_blocking=True)
    return out


def _host_collate(items):
    return collate_frames(items, device=None)


class _DeviceLoader:
    """The training loader: the first pass over the dataset comes from a DataLoader whose worker processes decode
    and collate on the host (pinned, uint8 frames); every sample that arrives is KEPT ON THE DEVICE — a 1024^2 frame
    is 3 MB as uint8, a thousand-frame capture 3 GB of the GPU's 288 GB — and once every frame has been seen the
    epochs are served from HBM: the DataLoader's own batch sampler still draws the (shuffled, rank-sharded) indices,
    but the batch is assembled on the device. No worker hand-over, no pinning thread, no PCIe, no per-epoch iterator
    restart (the reference's loop re-creates its iterator every epoch and decodes every PNG again,
    /root/reference/train.py:63, /root/reference/scene/dataset_mono.py:204-233: ~1-2 ms of waiting per 4 ms
    iteration, 20 ms at every epoch start). `budget_gb` bounds the resident set; a larger dataset keeps streaming."""

    def __init__(self, loader, device, budget_gb: Optional[float] = None):
        self.loader, self.device = loader, torch.device(device)
        if budget_gb is None:
            # a quarter of what is free on THIS device when training starts, at most 64 GB: a 288 GB part keeps the old
            # figure, a smaller or a busy one keeps streaming instead of running out of memory
            budget_gb = 64.0
            if self.device.type == "cuda":
                budget_gb = min(64.0, 0.25 * torch.cuda.mem_get_info(self.device)[0] / float(1 << 30))
        self.budget = int(budget_gb * (1 << 30))
        self.samples = {}           # dataset index -> {key: per-sample value on the device}
        self.bytes = 0
        self.streaming = False      # the dataset does not fit the budget: never serve from the cache

    def __len__(self):
        return len(self.loader)

    def _to_device(self, batch):
        return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    @staticmethod
    def _finish(out):
        img = out.get("original_image")
        if torch.is_tensor(img) and img.dtype == torch.uint8:      # dataset.raw_uint8: the reference's image / 255
            out["original_image"] = img.float().div_(255.0)
        return out

    def _keep(self, out):
        """Remember the samples of a device batch (views of its tensors: no copy)."""
        idx = out.get("pose_idx")
        if self.streaming or not torch.is_tensor(idx):
            return
        for b, i in enumerate(idx.tolist()):
            if i in self.samples:
                continue
            smp = {k: (v[b] if (torch.is_tensor(v) or isinstance(v, list)) else v) for k, v in out.items()}
            self.bytes += sum(v.numel() * v.element_size() for v in smp.values() if torch.is_tensor(v))
            if self.bytes > self.budget:
                self.samples.clear()
                self.streaming = True
                return
            self.samples[i] = smp

    def _from_cache(self, indices):
        smps = [self.samples[i] for i in indices]
        out = {}
        for k, v0 in smps[0].items():
            out[k] = torch.stack([s[k] for s in smps]) if torch.is_tensor(v0) else [s[k] for s in smps]
        return out

    def __iter__(self):
        n = len(self.loader.dataset)
        if not self.streaming and len(self.samples) == n and all(i in self.samples for i in range(n)):
            for indices in self.loader.batch_sampler:              # shuffling / rank sharding / drop_last as configured
                yield self._finish(self._from_cache(indices))
            return
        for batch in self.loader:
            out = self._to_device(batch)
            self._keep(out)
            yield self._finish(dict(out))


class AvatarModel:
    def __init__(self, model_parms, net_parms, opt_parms, load_iteration=None, train=True,
                 assets: Optional[dict] = None, frames: Optional[dict] = None, device="cuda"):
        self.model_parms, self.net_parms, self.opt_parms = model_parms, net_parms, opt_parms
        self.model_path = model_parms.model_path
        self.loaded_iter = None
        self.train = train
        self.train_mode = model_parms.train_mode
        self.device = torch.device(device)
        self.batch_size = model_parms.batch_size if train else 1
        assert model_parms.smpl_type in ("smplx", "smpl")
        S = model_parms.query_posmap_size
        split = "train" if train else "test"
        src = getattr(model_parms, "source_path", "")
        self.from_disk = assets is None and bool(src) and os.path.isdir(os.path.join(src, split))
        if self.from_disk:
            # the reference's layout (avatar_model.py:41-98): frames + pose table from the dataset,
            # masks / lbs map / posmaps / canonical joint transforms / body model from their files
            self.train_dataset = disk.MonoDataset_train(model_parms)
            self.smpl_data = self.train_dataset.smpl_data
            assets = disk.load_assets(model_parms, split)
            frames = dict(pose=self.train_dataset.pose_data.float(), transl=self.train_dataset.transl_data.float())
        else:
            if assets is None:
                assets = make_assets(getattr(model_parms, "num_points", 200_000), S, model_parms.smpl_type)
            if frames is None:
                frames = make_frames(assets, getattr(model_parms, "num_frames", 16),
                                     getattr(model_parms, "image_width", 1024),
                                     getattr(model_parms, "image_height", 1024))
            self.train_dataset = SyntheticFrames(frames, model_parms.train_stage, model_parms.inp_posmap_size)
        self.assets, self.frames = assets, frames
        dev = self.device
        if "fix_inp_map" in assets:
            self.fix_inp_map = assets["fix_inp_map"].to(dev)[None].expand(self.batch_size, -1, -1, -1)
        joint_num = assets["num_joints"]
        self.smpl_model = SMPLBody(assets["joints_rest"], assets["parents"]).to(dev).eval()
        valid = assets["valid_idx"].reshape(-1)
        self.valid_idx = valid.to(dev)
        self.valid_index = torch.nonzero(valid, as_tuple=False).reshape(-1).to(dev)      # int64 [N]
        inv = torch.full((valid.numel(),), -1, dtype=torch.int64)
        inv[self.valid_index.cpu()] = torch.arange(self.valid_index.numel())
        self.inv_index = inv.to(dev)                                                     # int64 [HW]
        self.uv_coord_map = assets["uv_coord_map"].to(dev)                               # [HW,2], no grad
        query_map = assets["query_posmap"].reshape(-1, 3)
        self.query_points = query_map[valid].to(dev).contiguous()[None].expand(self.batch_size, -1, -1)
        N = self.query_points.shape[1]
        # opacity and rotation are fixed (avatar_model.py:79-83)
        self.fix_opacity = torch.ones((N, 1), device=dev)
        rots = torch.zeros((N, 4), device=dev)
        rots[:, 0] = 1
        self.fix_rotation = rots
        lbs = assets["lbs_map"].reshape(S * S, joint_num)
        self.query_lbs = lbs[valid].to(dev).contiguous()[None].expand(self.batch_size, -1, -1)
        self.inv_mats = torch.linalg.inv(assets["cano_joint_mat"]).to(dev).contiguous().expand(self.batch_size, -1, -1, -1)
        self.betas = assets["betas"][0][None].expand(self.batch_size, -1).to(dev)
        self.pose = nn.Embedding(len(self.train_dataset), frames["pose"].shape[1],
                                 _weight=frames["pose"].clone(), sparse=True).to(dev)
        self.transl = nn.Embedding(len(self.train_dataset), 3, _weight=frames["transl"].clone(), sparse=True).to(dev)
        self.optimizer_pose = torch.optim.SparseAdam(list(self.pose.parameters()) + list(self.transl.parameters()), 5.0e-3)
        bg_color = [1, 1, 1] if model_parms.white_background else [0, 0, 0]
        self.background = torch.tensor(bg_color, dtype=torch.float32, device=dev)
        self.optimizer = None
        self.scheduler = None
        self.net_set(model_parms.train_stage)
        from . import rasterizer
        if self.device.type == "cuda":
            rasterizer.reset_capacity_history()      # new model = new scene

    # ------------------------------------------------------------------ construction
    def net_set(self, mode):
        assert mode in [0, 1, 2]
        np_ = self.net_parms
        self.net = POP_no_unet(c_geom=np_.c_geom, geom_layer_type=np_.geom_layer_type, nf=np_.nf,
                               hsize=np_.hsize, up_mode=np_.up_mode, use_dropout=bool(np_.use_dropout),
                               uv_feat_dim=2).to(self.device)
        S_in = self.model_parms.inp_posmap_size
        geo = torch.ones(1, np_.c_geom, S_in, S_in).normal_(mean=0.0, std=0.01).float().to(self.device)
        if geo.is_cuda and np_.c_geom == 64 and S_in % 64 == 0:
            # same values, channels-last in memory: the layout the hand-written convolution / up-sampling kernels read
            # and the one their input gradient comes back in (no layout copy each way, gradient contributions add
            # without strides). Only where those kernels apply (fused.geom_convs_supported): the vendor convolutions
            # pick other, non-reproducible algorithms for channels-last inputs
            geo = geo.contiguous(memory_format=torch.channels_last)
        self.geo_feature = nn.Parameter(geo.requires_grad_(True))
        if self.model_parms.train_stage == 2:
            self.pose_encoder = UnetNoCond5DS(input_nc=3, output_nc=np_.c_pose, nf=np_.nf,
                                              up_mode=np_.up_mode, use_dropout=False).to(self.device)
            if parallel.world_size() > 1 and self.device.type == "cuda":
                # the frames of a batch are spread over the ranks: BatchNorm statistics over the global batch,
                # as in the single-process reference (state-dict compatible)
                self.pose_encoder = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.pose_encoder, parallel.process_group())
        self.sync_replicas()

    def training_setup(self):
        o = self.opt_parms
        if self.model_parms.train_stage == 1:
            groups = [{"params": self.net.parameters(), "lr": o.lr_net},
                      {"params": self.geo_feature, "lr": o.lr_geomfeat}]
        else:
            groups = [{"params": self.net.parameters(), "lr": o.lr_net * 0.1},
                      {"params": self.pose_encoder.parameters(), "lr": o.lr_net}]
        # same update rule as the reference's torch.optim.Adam (model/avatar_model.py:152-161); on a HIP
        # device it is applied to every tensor of every group with one launch (optim.Adam, state-dict compatible)
        if self.geo_feature.is_cuda:
            from .optim import Adam
            self.optimizer = Adam(groups)
        else:
            self.optimizer = torch.optim.Adam(groups)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, o.sched_milestones, gamma=0.1)
        if self.device.type == "cuda":
            # Everything built so far (torch's modules, the body model, the dataset: several 100 k container objects)
            # lives as long as the model: take it out of the cyclic collector's scans. A full collection over it is a
            # 70-95 ms host stall every ~100 iterations of a young process (tools/step_times.py: stage 2, steps 24 and
            # 152), during which nothing is enqueued — 4-5 ms per iteration in a 20-step measurement that catches one.
            # (gc.freeze is process-wide: thaw what an earlier model froze first, so that only the CURRENT model's objects are
            # exempt and an earlier model's cyclic garbage — which may hold device tensors — is collected now; `close()`
            # thaws for good. ADVICE r04.)
            import gc
            gc.unfreeze()
            gc.collect()
            gc.freeze()
            self._gc_frozen = True

    def close(self):
        """Undo the process-wide side effects of training_setup (the cyclic collector's frozen generation) and drop the
        resident training samples."""
        if getattr(self, "_gc_frozen", False):
            import gc
            gc.unfreeze()
            self._gc_frozen = False
        ref = getattr(self, "_device_loader", None)
        loader = ref() if ref is not None else None
        if loader is not None:
            loader.samples.clear()
            loader.bytes = 0

    # ------------------------------------------------------------------ checkpoints
    def _ckpt_dir(self, iteration):
        return os.path.join(self.model_path, "net/iteration_{}".format(iteration))

    def save(self, iteration):
        path = self._ckpt_dir(iteration)
        os.makedirs(path, exist_ok=True)
        state = {"net": self.net.state_dict(), "geo_feature": self.geo_feature,
                 "pose": self.pose.state_dict(), "transl": self.transl.state_dict(),
                 "optimizer": self.optimizer.state_dict(), "scheduler": self.scheduler.state_dict()}
        if self.model_parms.train_stage == 1:
            torch.save(state, os.path.join(path, "net.pth"))
        else:
            state["pose_encoder"] = self.pose_encoder.state_dict()
            torch.save(state, os.path.join(path, "pose_encoder.pth"))

    def load(self, iteration, test=False):
        path = self._ckpt_dir(iteration)
        saved = torch.load(os.path.join(path, "net.pth"), map_location=self.device, weights_only=False)
        self.net.load_state_dict(saved["net"], strict=False)
        if self.model_parms.train_stage == 1:
            if not test:
                self.pose.load_state_dict(saved["pose"], strict=False)
                self.transl.load_state_dict(saved["transl"], strict=False)
            self.geo_feature.data[...] = saved["geo_feature"].data[...]
        if self.optimizer is not None:
            self.optimizer.load_state_dict(saved["optimizer"])
        if self.scheduler is not None:
            self.scheduler.load_state_dict(saved["scheduler"])
        self.sync_replicas()

    def stage_load(self, ckpt_path):
        saved = torch.load(os.path.join(ckpt_path, "net.pth"), map_location=self.device, weights_only=False)
        self.net.load_state_dict(saved["net"], strict=False)
        self.pose.load_state_dict(saved["pose"], strict=False)
        self.transl.load_state_dict(saved["transl"], strict=False)
        self.geo_feature.data[...] = saved["geo_feature"].data[...]
        self.sync_replicas()

    def stage2_load(self, epoch):
        path = os.path.join(self.model_parms.project_path, self.model_path, "net/iteration_{}".format(epoch))
        st = torch.load(os.path.join(path, "pose_encoder.pth"), map_location=self.device, weights_only=False)
        self.net.load_state_dict(st["net"], strict=False)
        self.pose.load_state_dict(st["pose"], strict=False)
        self.transl.load_state_dict(st["transl"], strict=False)
        self.geo_feature.data[...] = st["geo_feature"].data[...]
        self.pose_encoder.load_state_dict(st["pose_encoder"], strict=False)
        self.sync_replicas()

    # ------------------------------------------------------------------ data
    def getTrainDataloader(self):
        # data parallel: every rank draws the same permutation and keeps its own slice of it (frames are
        # the sharding unit, parallel.ShardedSampler); a single process shuffles like the reference does
        sampler = parallel.ShardedSampler(len(self.train_dataset)) if parallel.world_size() > 1 else None
        if self.from_disk:      # image decoding in worker processes (avatar_model.py:238-244)
            workers = int(getattr(self.model_parms, "num_workers", 4))
            self.train_dataset.raw_uint8 = True       # uint8 frames to the device, / 255 there (_DeviceLoader)
            loader = torch.utils.data.DataLoader(
                self.train_dataset, batch_size=self.batch_size, shuffle=sampler is None, sampler=sampler,
                num_workers=workers, drop_last=True, collate_fn=_host_collate,
                pin_memory=self.device.type == "cuda", persistent_workers=workers > 0)
            dl = _DeviceLoader(loader, self.device, getattr(self.model_parms, "device_cache_gb", None))
            self._device_loader = weakref.ref(dl)
            return dl
        return torch.utils.data.DataLoader(self.train_dataset, batch_size=self.batch_size, shuffle=sampler is None,
                                           sampler=sampler, num_workers=0, drop_last=True,
                                           collate_fn=lambda items: collate_frames(items, self.device))

    def sync_replicas(self):
        """Data parallel: all ranks take rank 0's parameters, BatchNorm buffers and pose tables (after
        construction and after every checkpoint load; a no-op on one rank)."""
        mods = [self.net, self.pose, self.transl] + ([self.pose_encoder] if hasattr(self, "pose_encoder") else [])
        parallel.broadcast_state(mods, [self.geo_feature.data])

    def _free_dataset(self):
        return SyntheticFrames(self.frames, self.model_parms.train_stage, self.model_parms.inp_posmap_size, test=True)

    def getTestDataset(self):
        self.test_dataset = disk.MonoDataset_test(self.model_parms) if self.from_disk else self._free_dataset()
        return self.test_dataset

    def getNovelposeDataset(self):
        self.novel_pose_dataset = disk.MonoDataset_novel_pose(self.model_parms) if self.from_disk \
            else self._free_dataset()
        return self.novel_pose_dataset

    def getNovelviewDataset(self):
        self.novel_view_dataset = disk.MonoDataset_novel_view(self.model_parms, joints_rest=self.assets["joints_rest"]) \
            if self.from_disk else self._free_dataset()
        return self.novel_view_dataset

    # ------------------------------------------------------------------ optimisation
    def _pose_opt_active(self, epoch):
        return self.model_parms.train_stage == 1 and epoch > self.opt_parms.pose_op_start_iter

    def zero_grad(self, epoch):
        self.optimizer.zero_grad()
        if self._pose_opt_active(epoch):
            self.optimizer_pose.zero_grad()
        elif self.model_parms.train_stage == 1:
            # The reference leaves the sparse pose/transl gradients un-zeroed until the pose optimiser
            # starts (/root/reference/model/avatar_model.py:258-263): they pile up in .grad (a sparse
            # add + periodic coalesce every iteration) and are thrown away by the first active
            # zero_grad. Dropping them right away gives the same training trajectory.
            self.pose.weight.grad = None
            self.transl.weight.grad = None

    def step(self, epoch):
        if self.model_parms.train_stage == 2:
            parallel.allreduce_param_grads(list(self.net.parameters()) + list(self.pose_encoder.parameters()))
        elif parallel.texel_sharding():
            # every rank back-propagated its slice of the UV map: the parameter gradients are partial sums
            parallel.allreduce_param_grads(list(self.net.parameters()) + [self.geo_feature], average=False)
        parallel.wait_overflow_flag()     # every rank skips the step if any rank dropped a frame's gradient
        pose_on = self._pose_opt_active(epoch)
        dropped = None
        if pose_on and self.device.type == "cuda":
            # the pose optimiser must drop an overflowed iteration too: read the flag before optim.Adam lowers it
            from . import rasterizer
            dropped = rasterizer.overflow_flag(self.device).clone()
        self.optimizer.step()
        if dropped is not None and not getattr(self.optimizer, "skip_on_overflow", False):
            from . import rasterizer
            rasterizer.clear_overflow_flag(self.device)      # (a main optimiser that does not lower the flag itself)
        self.scheduler.step()
        if pose_on:
            parallel.allgather_sparse_grads([self.pose.weight, self.transl.weight])
            self._pose_step(dropped)

    def _pose_step(self, dropped):
        """SparseAdam on the pose / translation rows of the batch — undone, row by row and without a host sync, when the
        iteration overflowed (`dropped` = the device flag, 1 = drop): a SparseAdam step on zero gradients would still
        decay both moments and apply a momentum-only update, so the rows the step touched — parameter and
        both moments — are put back with a device-side select. (What is not undone is SparseAdam's host-side `step`
        counter: the bias corrections of later steps run one step ahead, as in optim.Adam.)"""
        opt = self.optimizer_pose
        saved = []
        if dropped is not None:
            for p in (self.pose.weight, self.transl.weight):
                if p.grad is None or not p.grad.is_sparse:
                    continue
                p.grad = p.grad.coalesce()
                rows = p.grad._indices()[0]
                st = opt.state.get(p, {})
                saved.append((p, rows, p.data[rows].clone(),
                              {k: (st[k][rows].clone() if k in st else None) for k in ("exp_avg", "exp_avg_sq")}))
        opt.step()
        if saved:
            drop = dropped.bool()
            for p, rows, old, moments in saved:
                p.data[rows] = torch.where(drop, old, p.data[rows])
                st = opt.state.get(p, {})
                for k, m in moments.items():
                    if k in st:
                        st[k][rows] = torch.where(drop, m if m is not None else torch.zeros_like(st[k][rows]), st[k][rows])

    # ------------------------------------------------------------------ the hot path
    def _body(self, pose, transl, rest_pose):
        B = pose.shape[0]
        inv = self.inv_mats[:B] if self.inv_mats.shape[0] >= B else self.inv_mats[:1].expand(B, -1, -1, -1)
        if self.model_parms.smpl_type == "smplx":
            return self.smpl_model.forward(
                betas=self.betas, global_orient=pose[:, :3], transl=transl, body_pose=pose[:, 3:66],
                jaw_pose=rest_pose[:, :3], leye_pose=rest_pose[:, 3:6], reye_pose=rest_pose[:, 6:9],
                left_hand_pose=rest_pose[:, 9:54], right_hand_pose=rest_pose[:, 54:], inv_mats=inv)
        # SMPL: the embedding row is global_orient | body_pose already (avatar_model.py:280-285)
        return self.smpl_model.forward(betas=self.betas, transl=transl, full_pose=pose, inv_mats=inv)

    def _decode(self, B, pose_featmap, iteration, warmup: bool):
        """Net -> per-Gaussian residuals/scales/colours on the valid texels.
        Returns (offset_loss = mean((0.02 pred_res)^2) over all texels, scale_loss = mean(scales),
        point_res [B,N,3], scales [B,N,3], colours [B,N,3])."""
        # geo_feature and the uv map are batch-invariant: they enter with batch size 1 (the convs run
        # once; in stage 2 the pose features broadcast against them), and in stage 1 the whole net runs
        # once for the batch (duplicating rows does not change BatchNorm's batch statistics)
        uv = self.uv_coord_map[None]
        scale_mult = 1e-3 * iteration if (warmup and iteration < 1000) else 1.0
        N = self.valid_index.shape[0]
        if self.geo_feature.is_cuda and pose_featmap is None and parallel.texel_sharding():
            return self._decode_texel_sharded(B, uv, scale_mult)
        if self.geo_feature.is_cuda:
            # decoder heads (logits) -> per-Gaussian records + both regulariser means in one kernel.
            # Stage 2 under data parallelism: this rank's frames are a share of the global batch's decoder
            # rows; BatchNorm statistics are synchronised (the single-process batch's semantics)
            mg = None
            if pose_featmap is not None and parallel.world_size() > 1:
                mg = pose_featmap.shape[0] * self.uv_coord_map.shape[0] * parallel.world_size()
            pending = getattr(self, "_pose_pending", None)
            if pending is not None:
                # the encoder's result comes from its own stream (_pose_features); network.forward_points waits for the event
                # the TENSOR carries, as late as it can. A tensor that lost the attribute on the way here (a view, a copy)
                # is waited for now instead: correctness first, the overlap is what is lost
                if pose_featmap is not None and getattr(pose_featmap, "_ga_ready", None) is None:
                    torch.cuda.current_stream(self.device).wait_event(pending)
                self._pose_pending = None
            res, s_logit, c_logit = self.net.forward_points(pose_featmap, self.geo_feature, uv, raw_heads=True,
                                                            m_global=mg)
            b = res.shape[0]
            flat, offset_loss, scale_loss = fused.decode_pack(res, s_logit, c_logit, self.valid_index,
                                                              self.inv_index, 0.02, scale_mult)
        else:
            res, scales, shs = self.net.forward_points(pose_featmap, self.geo_feature, uv)
            b = res.shape[0]
            res = res * 0.02
            scales = scales * scale_mult if scale_mult != 1.0 else scales
            offset_loss = torch.mean(res ** 2)
            pick = lambda t: t.index_select(1, self.valid_index)
            scale_loss = torch.mean(pick(scales))
            flat = torch.cat([pick(res).reshape(-1), pick(scales).reshape(-1), pick(shs).reshape(-1)])
        shared = b == 1 and B > 1
        if shared or self.model_parms.train_stage == 1:
            # one flat tensor (residual | scale | colour segments) is the DP exchange point
            # (parallel.exchange_output_grads is the identity on one rank)
            flat = parallel.exchange_output_grads(flat)
        point_res, scale3, pshs = fused.expand_records(flat, b, N, B if shared else b)
        return offset_loss, scale_loss, point_res, scale3, pshs

    def _decode_texel_sharded(self, B, uv, scale_mult):
        """Stage 1 with parallel.set_mode("texels"): this rank evaluates the decoder on its slice of the
        UV map only (BatchNorm statistics synchronised over ranks), packs its valid texels, and the full
        [N,7] record buffer is assembled on every rank with one all-reduce. Returned losses carry the global
        values; their gradients reach this rank's slice only (parameter gradients are summed in step())."""
        HW = self.uv_coord_map.shape[0]
        N = self.valid_index.shape[0]
        r0, r1 = parallel.shard_range(HW)
        if getattr(self, "_shard_key", None) != (r0, r1):
            vi = self.valid_index
            sel = (vi >= r0) & (vi < r1)
            n_local = int(sel.sum())
            n0 = int((vi < r0).sum())
            inv = self.inv_index[r0:r1]
            self._shard = dict(n0=n0, n_local=n_local, valid=(vi[sel] - r0).contiguous(),
                               inv=torch.where(inv >= 0, inv - n0, inv).contiguous())
            self._shard_key = (r0, r1)
        sh = self._shard
        res, s_logit, c_logit = self.net.forward_points(None, self.geo_feature, uv, raw_heads=True,
                                                        rows=(r0, r1), m_global=HW)
        flat_l, off_l, scale_l = fused.decode_pack(res, s_logit, c_logit, sh["valid"], sh["inv"], 0.02, scale_mult)
        # the local means are over this slice: weight them into the global means
        offset_loss = parallel.sum_over_ranks(off_l * ((r1 - r0) / float(HW)))
        scale_loss = parallel.sum_over_ranks(scale_l * (sh["n_local"] / float(N)))
        flat = parallel.gather_segments(flat_l, sh["n0"], sh["n_local"], N)
        point_res, scale3, pshs = fused.expand_records(flat, 1, N, B)
        return offset_loss, scale_loss, point_res, scale3, pshs

    def _render_frames(self, batch_data, full_pred, colors, scales):
        """The reference renders the frames one by one (avatar_model.py:332-365); here the whole
        batch goes through one launch of every rasterizer kernel."""
        B = full_pred.shape[0]
        return render_frames(
            points=full_pred, colors_precomp=colors, rotations=self.fix_rotation, scales=scales,
            opacity=self.fix_opacity, FovX=batch_data["FovX"], FovY=batch_data["FovY"],
            height=batch_data["height"], width=batch_data["width"], bg_color=self.background,
            world_view_transform=batch_data["world_view_transform"][:B],
            full_proj_transform=batch_data["full_proj_transform"][:B],
            camera_center=batch_data["camera_center"][:B])

    def _forward(self, batch_data, iteration, pose, transl, pose_featmap, warmup):
        B = pose.shape[0]
        # the decoder first: its launches are long, so in a loop that drains the GPU every iteration (the reference's
        # train.py reads loss.item()) the host-bound launches of the body model run behind them instead of in front of an
        # idle GPU (~40-75 us per iteration there; nothing changes when the host runs ahead)
        offset_loss, scale_loss, point_res, scales, colors = self._decode(B, pose_featmap, iteration, warmup)
        live = self._body(pose, transl, batch_data.get("rest_pose"))
        full_pred = skin(self.query_points[:B] if self.query_points.shape[0] >= B else self.query_points[:1].expand(B, -1, -1),
                         point_res, self.query_lbs[0], live.cano2live)
        image = self._render_frames(batch_data, full_pred, colors, scales)
        return image, full_pred, offset_loss, scale_loss

    def train_stage1(self, batch_data, iteration):
        idx = batch_data["pose_idx"]
        image, full_pred, offset_loss, scale_loss = self._forward(
            batch_data, iteration, self.pose(idx), self.transl(idx), None, warmup=True)
        geo_loss = fused.mean_sq(self.geo_feature) if self.geo_feature.is_cuda else torch.mean(self.geo_feature ** 2)
        if parallel.texel_sharding():
            geo_loss = parallel.replicated_term(geo_loss)       # every rank computes it; gradients are summed
        return image, full_pred, offset_loss, geo_loss, scale_loss

    def _pose_features(self, inp):
        """The pose encoder's feature map. On a HIP device the encoder runs on a stream of its own: its ~30 launches are a
        serial chain of kernels that fill a fraction of the chip (csrc/ganet_unet.hip), and nothing else of the forward
        pass needs its result before the feature maps are added (network.forward_points waits for the event left on the
        tensor) — the geometry net's convolutions and the body model run beside it, and so do their backward passes
        beside the encoder's (autograd runs a node's backward on the stream of its forward and orders the streams)."""
        # (one rank only: with several, the encoder's SyncBatchNorm issues collectives, which stay on the main stream)
        if not (inp.is_cuda and _dev.knobs.encoder_stream and torch.is_grad_enabled() and parallel.world_size() == 1):
            return self.pose_encoder(inp)
        cur = torch.cuda.current_stream(inp.device)
        side = fused.encoder_stream(inp.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            feat = self.pose_encoder(inp)
            ready = fused.encoder_event(inp.device)
            ready.record(side)
        inp.record_stream(side)
        feat.record_stream(cur)
        feat._ga_ready = ready
        self._pose_pending = ready          # (the fallback wait of _decode, should the attribute get lost on the way)
        return feat

    def train_stage2(self, batch_data, iteration):
        idx = batch_data["pose_idx"]
        pose_featmap = self._pose_features(batch_data["inp_pos_map"])
        image, full_pred, offset_loss, _scale_loss = self._forward(
            batch_data, iteration, self.pose(idx), self.transl(idx), pose_featmap, warmup=False)
        # (one launch each way instead of pow / mean / their three backward kernels over the 8 MB feature map)
        pose_loss = fused.mean_sq(pose_featmap) if pose_featmap.is_cuda else torch.mean(pose_featmap ** 2)
        return image, full_pred, pose_loss, offset_loss,

    def render_free_stage1(self, batch_data, iteration):
        image, _, _, _ = self._forward(batch_data, iteration, batch_data["pose_data"],
                                       batch_data["transl_data"], None, warmup=True)
        return image

    def render_free_stage2(self, batch_data, iteration):
        # the reference looks the pose up in the learned embeddings here (model/avatar_model.py:555-560),
        # unlike render_free_stage1 which takes batch['pose_data']
        idx = batch_data["pose_idx"]
        pose_featmap = self.pose_encoder(batch_data["inp_pos_map"])
        image, _, _, _ = self._forward(batch_data, iteration, self.pose(idx), self.transl(idx),
                                       pose_featmap, warmup=False)
        return image
