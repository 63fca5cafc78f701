"""Data layer: readers for the reference's ON-DISK formats, either side of the hot path.

What the reference loads, and from where:

    <source_path>/<split>/images/<name>.<ext>          RGB frames            dataset_mono.py:126-133,208
    <source_path>/<split>/masks/<name>.<ext>           foreground masks      dataset_mono.py:135,214-222
    <source_path>/<split>/cam_parms/<name>.npz         per-frame camera      dataset_mono.py:191-197
    <source_path>/<split>/cam_parms.npz                static camera         dataset_mono.py:159-166
    <source_path>/<split>/smpl_parms.pth               {beta, trans, body_pose}   (stage 1)
    <source_path>/<split>/smpl_parms_pred.pth          same, optimised poses      (stage 2)
    <source_path>/<split>/inp_map/inp_posemap_<S>_<idx:08d>.npz     key posmap<S>  (stage 2)
    <source_path>/<split>/query_posemap_<S>_cano_<smpl>.npz         key posmap<S>  avatar_model.py:52,66
    <source_path>/<split>/<smpl>_cano_joint_mat.pth    canonical joint transforms [J,4,4]  avatar_model.py:56,89
    <project_path>/assets/uv_masks/uv_mask<S>_with_faceid_<smpl>.npy   face id per texel, -1 = empty
    <project_path>/assets/lbs_map_<smpl>_<S>.npy       skinning weights per texel [S,S,J]
    <smpl_model_path>/SMPL_<GENDER>.pkl | <smplx_model_path>/SMPLX_<GENDER>.npz   body model files

The classes keep the reference's names and item keys so its train.py / eval.py /
render_novel_pose.py loops (default DataLoader collate + `to_cuda`) run on them unchanged.
`synthetic.write_dataset` produces this layout from the seeded synthetic assets, because none of
it ships with the reference. Only PIL + numpy are needed (the reference also imports cv2 for one
Rodrigues call — restated below).
"""
from __future__ import annotations

import collections
import os
import pickle
from os.path import join

import numpy as np
import torch
from torch.utils.data import Dataset

from .camera import make_camera, projection_matrix


# ----------------------------------------------------------------------------- helpers
def to_cuda(items: dict, device, add_batch: bool = False, precision=torch.float32):
    """utils/general_utils.py:132-163: tensors / arrays (also one level inside dict values) go
    to `device`, floating tensors are cast to `precision`, everything else passes through."""
    def move(v):
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        if isinstance(v, torch.Tensor):
            v = v.to(device)
            if v.dtype in (torch.float32, torch.float64):
                v = v.to(precision)
        return v

    out = {}
    for key, data in items.items():
        if isinstance(data, dict):
            inner = {}
            for k2, d2 in data.items():
                if not isinstance(d2, (np.ndarray, torch.Tensor)):
                    raise TypeError("Do not support other data types.")
                inner[k2] = move(d2)
            data = inner
        else:
            data = move(data)
        if add_batch:
            if isinstance(data, torch.Tensor):
                data = data.unsqueeze(0)
            elif isinstance(data, dict):
                data = {k: v.unsqueeze(0) for k, v in data.items()}
            else:
                data = [data]
        out[key] = data
    return out


def uv_index_map(size: int) -> torch.Tensor:
    """getIdxMap_torch without offset (utils/general_utils.py:165-176): [S*S,2] = (row, col)/(S-1)."""
    r, c = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    return torch.stack([r.reshape(-1), c.reshape(-1)], dim=1).float() / (size - 1)


def load_masks(project_path: str, posmap_size: int, body_model: str = "smpl"):
    """utils/general_utils.py:178-191 -> (flist_uv, valid_idx bool[S*S], uv_coord_map [S*S,2]).
    `flist_uv` (vertex triple of the face under each valid texel) is returned when
    assets/<body_model>_faces.npy exists and is None otherwise — the render-and-fit path never
    reads it. uv_coord_map carries no grad."""
    S = posmap_size
    faceid = np.load(join(project_path, "assets", "uv_masks",
                          "uv_mask{}_with_faceid_{}.npy".format(S, body_model))).reshape(S, S)
    faceid = torch.from_numpy(faceid.astype(np.int64))
    valid = (faceid != -1).reshape(-1)
    flist_uv = None
    faces_path = join(project_path, "assets", "{}_faces.npy".format(body_model.lower()))
    if os.path.exists(faces_path):
        flist = torch.from_numpy(np.load(faces_path).astype(np.int64))
        flist_uv = flist[faceid.reshape(-1)[valid]]
    return flist_uv, valid, uv_index_map(S)


class _ChumpyStub:
    """Stand-in for chumpy.Ch objects inside the official SMPL pickles (chumpy is not a
    dependency here): keeps the pickled state, `.r` is the array."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"x": state})

    @property
    def r(self):
        return np.asarray(self.__dict__.get("x"))


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "chumpy":
            return _ChumpyStub
        return super().find_class(module, name)


def _dense(a):
    if isinstance(a, _ChumpyStub):
        a = a.r
    if hasattr(a, "toarray"):          # scipy.sparse J_regressor of the SMPL pickles
        a = a.toarray()
    return np.asarray(a)


def load_body_model(model_path: str, smpl_type: str = "smpl", gender: str = "neutral") -> dict:
    """The fields of an SMPL / SMPL-X model file that the joint path needs
    (submodules/smplx/body_models.py:127-138 for SMPL .pkl, :966-981 for SMPL-X .npz/.pkl):
    v_template [V,3], shapedirs [V,3,>=10], J_regressor [J,V], parents [J] (kintree_table[0])."""
    if os.path.isdir(model_path):
        stem = ("SMPL_{}" if smpl_type == "smpl" else "SMPLX_{}").format(gender.upper())
        cands = [join(model_path, stem + e) for e in ((".pkl", ".npz") if smpl_type == "smpl" else (".npz", ".pkl"))]
        path = next((c for c in cands if os.path.exists(c)), None)
        assert path is not None, "Path {} does not exist!".format(cands[0])
    else:
        path = model_path
        assert os.path.exists(path), "Path {} does not exist!".format(path)
    if path.endswith(".npz"):
        data = dict(np.load(path, allow_pickle=True))
    else:
        with open(path, "rb") as f:
            data = _TolerantUnpickler(f, encoding="latin1").load()
    parents = _dense(data["kintree_table"])[0].astype(np.int64).copy()
    parents[0] = -1
    return dict(v_template=torch.tensor(_dense(data["v_template"]), dtype=torch.float32),
                shapedirs=torch.tensor(_dense(data["shapedirs"]), dtype=torch.float32),
                J_regressor=torch.tensor(_dense(data["J_regressor"]), dtype=torch.float32),
                parents=parents.astype(np.int32))


def rest_joints(body: dict, betas) -> torch.Tensor:
    """J(betas) = J_regressor (v_template + shapedirs[..., :len(betas)] betas)   (lbs.py:206-210)."""
    betas = torch.as_tensor(betas, dtype=torch.float32).reshape(-1)
    v = body["v_template"] + torch.einsum("l,mkl->mk", betas, body["shapedirs"][:, :, :betas.numel()])
    return torch.einsum("ik,ji->jk", v, body["J_regressor"])


def _load_smpl_parms(path):
    d = torch.load(path, map_location="cpu", weights_only=False)
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}


def load_assets(model_parms, split: str) -> dict:
    """Everything AvatarModel.__init__ reads besides the frames (avatar_model.py:44-98), as the
    `assets` dict the model is built from."""
    st = model_parms.smpl_type
    S = model_parms.query_posmap_size
    folder = join(model_parms.source_path, split)
    _flist, valid, uv = load_masks(model_parms.project_path, S, body_model=st)
    qmap = np.load(join(folder, "query_posemap_{}_cano_{}.npz".format(S, st)))["posmap" + str(S)]
    lbs = np.load(join(model_parms.project_path, "assets", "lbs_map_{}_{}.npy".format(st, S)))
    mats = torch.load(join(folder, "{}_cano_joint_mat.pth".format(st)), map_location="cpu", weights_only=False)
    mats = torch.as_tensor(mats, dtype=torch.float32)
    nj = 55 if st == "smplx" else 24
    smpl_file = "smpl_parms.pth" if model_parms.train_stage == 1 else "smpl_parms_pred.pth"
    parms = _load_smpl_parms(join(folder, smpl_file))
    body = load_body_model(model_parms.smplx_model_path if st == "smplx" else model_parms.smpl_model_path,
                           st, model_parms.smpl_gender)
    assert body["parents"].shape[0] == nj, (body["parents"].shape, nj)
    betas = parms["beta"].float().reshape(-1, parms["beta"].shape[-1])
    assets = dict(smpl_type=st, num_joints=nj, parents=body["parents"],
                  joints_rest=rest_joints(body, betas[0][:10]),
                  valid_idx=valid, uv_coord_map=uv,
                  query_posmap=torch.from_numpy(np.asarray(qmap, np.float32)).reshape(S, S, 3),
                  lbs_map=torch.from_numpy(np.asarray(lbs, np.float32)).reshape(S, S, nj),
                  cano_joint_mat=mats.reshape(1, nj, 4, 4), betas=betas)
    if getattr(model_parms, "fixed_inp", 0):
        Si = model_parms.inp_posmap_size
        inp = np.load(join(folder, "query_posemap_{}_cano_{}.npz".format(Si, st)))["posmap" + str(Si)]
        assets["fix_inp_map"] = torch.from_numpy(np.asarray(inp, np.float32).transpose(2, 0, 1))
    return assets


def _rodrigues(v):
    """cv2.Rodrigues(v)[0] for a 3-vector."""
    v = np.asarray(v, np.float64)
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def rotate_camera_by_frame_idx(extrinsics, frame_idx, trans=None, rotate_axis="y", period=196, inv_angle=False):
    """Orbit the camera about a world axis through `trans` (dataset_mono.py:10-95)."""
    angle = 2 * np.pi * (frame_idx / period)
    if inv_angle:
        angle = -angle
    inv_E = np.linalg.inv(np.asarray(extrinsics, np.float64))
    camrot, campos = inv_E[:3, :3], inv_E[:3, 3].copy()
    if trans is not None:
        campos -= trans
    if camrot.T[1, 1] < 0.0:
        angle = -angle
    vec = np.zeros(3)
    vec[{"x": 0, "y": 1, "z": 2}[rotate_axis]] = angle
    g = _rodrigues(vec).astype(np.float32).astype(np.float64)
    pos, rot = g @ campos, g @ camrot
    if trans is not None:
        pos += trans
    E = np.identity(4)
    E[:3, :3] = rot.T
    E[:3, 3] = -rot.T @ pos
    return E


# ----------------------------------------------------------------------------- datasets
class _MonoBase(Dataset):
    """Shared reader. Subclasses set the folder, which SMPL file to read, whether frames come
    from images/ (train/test/novel view) or from the pose table alone (novel pose)."""

    with_images = True
    with_pose = False          # items carry pose_data / transl_data (all but the training set)
    # Training-loader fast path (AvatarModel.getTrainDataloader sets it): `original_image` leaves the worker as the
    # composited uint8 [3,H,W] — a quarter of the bytes through shared memory, pinning and PCIe — and becomes the
    # reference's float image / 255 on the device (avatar_model._DeviceLoader). Items read directly keep the
    # reference's format (float [3,H,W] in [0,1]).
    raw_uint8 = False
    CACHE_MB = 1024            # decoded (mask-composited) uint8 frames kept per process, least recently used first out

    def __init__(self, dataset_parms, folder, device=torch.device("cuda:0"), predicted_poses=None):
        super().__init__()
        p = self.dataset_parms = dataset_parms
        self.data_folder, self.device = folder, device
        self.gender = p.smpl_gender
        self.zfar, self.znear = 100.0, 0.01
        self.trans, self.scale = np.array([0.0, 0.0, 0.0]), 1.0
        self.no_mask = bool(p.no_mask)
        if predicted_poses is None:
            predicted_poses = p.train_stage != 1
        self.smpl_data = _load_smpl_parms(join(folder, "smpl_parms_pred.pth" if predicted_poses else "smpl_parms.pth"))
        if self.with_images:
            files = sorted(os.listdir(join(folder, "images")))
            self.data_length = len(files)
            self.name_list = [(i, f.split(".")[0]) for i, f in enumerate(files)]
            self.image_fix = files[0].split(".")[-1]
            if not self.no_mask:
                self.mask_fix = sorted(os.listdir(join(folder, "masks")))[0].split(".")[-1]
        else:
            self.data_length = self.smpl_data["body_pose"].shape[0]
        bp, tr = self.smpl_data["body_pose"].float(), self.smpl_data["trans"].float()
        n = self.data_length if self.with_images else bp.shape[0]
        if p.smpl_type == "smplx":
            self.pose_data, self.rest_pose_data = bp[:n, :66], bp[:n, 66:]
        else:
            self.pose_data = bp[:n]
        self.transl_data = tr[:n]
        if p.cam_static:
            cam = np.load(join(folder, "cam_parms.npz"))
            self.extr_npy = np.asarray(cam["extrinsic"], np.float32)
            self.intrinsic = np.asarray(cam["intrinsic"], np.float32).reshape(3, 3)

    def __len__(self):
        return self.data_length

    def __getitem__(self, index, ignore_list=None):
        return self.getitem(index, ignore_list)

    # -- pieces
    def _camera(self, name_idx):
        if self.dataset_parms.cam_static:
            return self.extr_npy, self.intrinsic
        cam = np.load(join(self.data_folder, "cam_parms", name_idx + ".npz"))
        return np.asarray(cam["extrinsic"], np.float32), np.asarray(cam["intrinsic"], np.float32).reshape(3, 3)

    def _image_u8(self, name_idx):
        """The frame as uint8 [H,W,C], background set to white through the mask (dataset_mono.py:208-233: mask < 128
        -> 0, else 1; image*mask + (1-mask)*255). Decoding a 1024^2 PNG pair costs ~10-20 ms of a worker's time — more
        than a whole training iteration on the GPU — so decoded frames are kept (per process, CACHE_MB, LRU): the
        reference decodes every frame again in every epoch."""
        cache = self.__dict__.setdefault("_frames", collections.OrderedDict())
        hit = cache.get(name_idx)
        if hit is not None:
            cache.move_to_end(name_idx)
            return hit
        from PIL import Image
        image = np.array(Image.open(join(self.data_folder, "images", name_idx + "." + self.image_fix)))
        if not self.no_mask:
            mask = np.array(Image.open(join(self.data_folder, "masks", name_idx + "." + self.mask_fix)))
            if mask.ndim < 3:
                mask = mask[..., None]
            fg = mask >= 128
            image = np.where(fg, image if image.ndim == 3 else image[..., None], 255).astype(np.uint8)
        if image.ndim < 3:
            image = image[..., None]
        budget = int(getattr(self.dataset_parms, "cache_mb", self.CACHE_MB)) << 20
        if image.nbytes <= budget:
            cache[name_idx] = image
            self._frames_bytes = self.__dict__.get("_frames_bytes", 0) + image.nbytes
            while self._frames_bytes > budget:
                _, old = cache.popitem(last=False)
                self._frames_bytes -= old.nbytes
        return image

    def _image(self, name_idx):
        """RGB in [0,1] as [3,H,W] float (raw_uint8: the same pixels as uint8 [3,H,W], divided by 255 on the device)."""
        image = self._image_u8(name_idx)
        if self.raw_uint8:
            return torch.from_numpy(np.ascontiguousarray(image.transpose(2, 0, 1)))
        t = torch.from_numpy(image) / 255.0
        return t.permute(2, 0, 1).clamp(0.0, 1.0)

    def _inp_posmap(self, pose_idx):
        S = self.dataset_parms.inp_posmap_size
        path = self.data_folder + "/inp_map/" + "inp_posemap_%s_%s.npz" % (str(S), str(pose_idx).zfill(8))
        return np.load(path)["posmap" + str(S)].transpose(2, 0, 1)

    def _item(self, pose_idx, extr, intrinsic, width, height, image=None):
        cam = make_camera(intrinsic, extr, width, height, self.znear, self.zfar)
        item = {}
        if self.dataset_parms.train_stage == 2:
            item["inp_pos_map"] = self._inp_posmap(pose_idx)
        if image is not None:
            item["original_image"] = image
        item.update(FovX=cam["FovX"], FovY=cam["FovY"], width=width, height=height, pose_idx=pose_idx)
        if self.with_pose:
            item["pose_data"] = self.pose_data[pose_idx]
            item["transl_data"] = self.transl_data[pose_idx]
        if self.dataset_parms.smpl_type == "smplx":
            item["rest_pose"] = self.rest_pose_data[pose_idx]
        wvt = torch.from_numpy(cam["world_view_transform"])
        full = torch.from_numpy(cam["full_proj_transform"])
        item["world_view_transform"] = wvt
        item["projection_matrix"] = torch.from_numpy(np.ascontiguousarray(projection_matrix(
            self.znear, self.zfar, cam["FovX"], cam["FovY"], np.asarray(intrinsic, np.float64), height, width).T))
        item["full_proj_transform"] = full
        item["camera_center"] = torch.from_numpy(cam["camera_center"])
        return item

    @torch.no_grad()
    def getitem(self, index, ignore_list=None):
        pose_idx, name_idx = self.name_list[index]
        extr, intrinsic = self._camera(name_idx)
        image = self._image(name_idx)
        return self._item(pose_idx, extr, intrinsic, image.shape[2], image.shape[1], image)


class MonoDataset_train(_MonoBase):
    """scene/dataset_mono.py:98-257 — <source_path>/train."""

    def __init__(self, dataset_parms, device=torch.device("cuda:0")):
        super().__init__(dataset_parms, join(dataset_parms.source_path, "train"), device)


class MonoDataset_test(_MonoBase):
    """scene/dataset_mono.py:259-417 — <source_path>/test; items also carry the frame's pose."""
    with_pose = True

    def __init__(self, dataset_parms, device=torch.device("cuda:0")):
        super().__init__(dataset_parms, join(dataset_parms.source_path, "test"), device)


class MonoDataset_novel_pose(_MonoBase):
    """scene/dataset_mono.py:419-522 — a pose table (`test_folder`/smpl_parms.pth) seen through
    the static camera of that folder at 1024 x 1024; no images."""
    with_images = False
    with_pose = True

    def __init__(self, dataset_parms, device=torch.device("cuda:0")):
        super().__init__(dataset_parms, dataset_parms.test_folder, device, predicted_poses=False)

    @torch.no_grad()
    def getitem(self, index, ignore_list=None):
        return self._item(index, self.extr_npy, self.intrinsic, 1024, 1024)


class MonoDataset_novel_view(_MonoBase):
    """scene/dataset_mono.py:524-672 — one fixed pose of the test split, camera orbiting the
    pelvis. `update_smpl(pose_idx, frame_num)` must be called first; the pelvis comes from
    `joints_rest` (rest joints of the body model, e.g. assets['joints_rest']) instead of the
    reference's numpy SMPL copy."""
    with_pose = True
    ROT_CAM_PARAMS = {"zju_mocap": {"rotate_axis": "z", "inv_angle": True},
                      "wild": {"rotate_axis": "y", "inv_angle": False}}

    def __init__(self, dataset_parms, device=torch.device("cuda:0"), joints_rest=None):
        super().__init__(dataset_parms, join(dataset_parms.source_path, "test"), device)
        self.src_type = "wild"
        self.joints_rest = joints_rest

    def update_smpl(self, pose_idx, frame_num):
        if self.joints_rest is None:
            p = self.dataset_parms
            body = load_body_model(p.smplx_model_path if p.smpl_type == "smplx" else p.smpl_model_path,
                                   p.smpl_type, p.smpl_gender)
            self.joints_rest = rest_joints(body, self.smpl_data["beta"].reshape(-1)[:10])
        pelvis = np.asarray(self.joints_rest[0], np.float64)
        self.Th = pelvis + self.smpl_data["trans"][pose_idx].double().numpy()
        self.data_length = frame_num
        self.fix_pose_idx = pose_idx

    def get_freeview_camera(self, frame_idx, total_frames, trans):
        return rotate_camera_by_frame_idx(extrinsics=self.extr_npy, frame_idx=frame_idx, period=total_frames,
                                          trans=trans, **self.ROT_CAM_PARAMS[self.src_type])

    @torch.no_grad()
    def getitem(self, index, ignore_list=None):
        from PIL import Image
        _, name_idx = self.name_list[0]
        with Image.open(join(self.data_folder, "images", name_idx + "." + self.image_fix)) as im:
            width, height = im.size
        extr = self.get_freeview_camera(index, self.data_length, self.Th).astype(np.float32)
        return self._item(self.fix_pose_idx, extr, self.intrinsic, width, height)
