"""Deterministic synthetic stand-ins for everything the reference loads from disk.

Nothing the hot path needs at run time ships with the reference (SMPL model files are
licence-gated, assets.zip/datasets/checkpoints are absent — SURVEY.md §0 fact 5), so the
benchmark configurations of BASELINE.json run on seeded synthetic assets with the shapes,
dtypes and conventions of the real ones (SURVEY.md §8d "Synthetic body model"):

  body model        SMPL kinematic tree (24 joints) or an SMPL-X-shaped 55-joint tree on a
                    1.7 m humanoid rest skeleton
  canonical points  N points on capsules around the bones, posed into the reference's canonical
                    pose (legs +-30 deg, arguments/__init__.py:44-53; root transl (0,0.3,0),
                    scripts/gen_pose_map_cano_smpl.py:62)
  query_lbs         softmax(-d^2/sigma^2) over bone distances, top-4, renormalised, dense [N,J]
  valid_idx         S x S UV mask with exactly N valid texels (uv_mask*_with_faceid, load_masks)
  uv_coord_map      getIdxMap_torch: (row, col)/(S-1)   (utils/general_utils.py:165-176)
  inv_mats          inverse of the canonical-pose joint transforms (smpl_cano_joint_mat.pth)
  poses/camera      the 16 poses + the pinhole camera the reference ships under
                    assets/test_pose (copied to gaussianavatar_amd/assets/test_pose.npz)
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from .camera import make_camera

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]

# T-pose rest joints of a 1.7 m humanoid, y up, pelvis near SMPL's (0,-0.22,0.03).
_SMPL_REST = np.array([
    [0.00, 0.00, 0.00], [0.07, -0.09, 0.0], [-0.07, -0.09, 0.0], [0.0, 0.11, 0.0],
    [0.10, -0.47, 0.0], [-0.10, -0.47, 0.0], [0.0, 0.25, 0.0], [0.09, -0.87, -0.03],
    [-0.09, -0.87, -0.03], [0.0, 0.30, 0.0], [0.11, -0.93, 0.09], [-0.11, -0.93, 0.09],
    [0.0, 0.51, 0.0], [0.08, 0.42, 0.0], [-0.08, 0.42, 0.0], [0.0, 0.60, 0.03],
    [0.18, 0.45, 0.0], [-0.18, 0.45, 0.0], [0.44, 0.45, 0.0], [-0.44, 0.45, 0.0],
    [0.69, 0.45, 0.0], [-0.69, 0.45, 0.0], [0.78, 0.45, 0.0], [-0.78, 0.45, 0.0]],
    dtype=np.float64) + np.array([0.0, -0.22, 0.03])

_RADIUS = {0: 0.13, 3: 0.13, 6: 0.13, 9: 0.12, 12: 0.055, 15: 0.10, 1: 0.075, 2: 0.075, 4: 0.055,
           5: 0.055, 7: 0.04, 8: 0.04, 10: 0.035, 11: 0.035, 13: 0.06, 14: 0.06, 16: 0.05, 17: 0.05,
           18: 0.04, 19: 0.04, 20: 0.03, 21: 0.03, 22: 0.025, 23: 0.025}


def _load_test_pose():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "test_pose.npz"))


def smplx_like_tree(seed: int = 0):
    """55-joint tree: the 22 SMPL body joints (without the two hand joints), jaw + 2 eyes on the
    head, 15 + 15 finger joints as three-link chains on the wrists."""
    parents = SMPL_PARENTS[:22] + [15, 15, 15]
    rest = list(_SMPL_REST[:22]) + [_SMPL_REST[15] + d for d in ([0, -0.05, 0.06], [0.03, 0.04, 0.08], [-0.03, 0.04, 0.08])]
    for side, wrist in ((1.0, 20), (-1.0, 21)):
        for f in range(5):
            base = len(parents)
            for k in range(3):
                parents.append(wrist if k == 0 else base + k - 1)
                rest.append(_SMPL_REST[wrist] + np.array([side * (0.05 + 0.03 * k), 0.0, 0.02 * (f - 2)]))
    return parents, np.asarray(rest, np.float64)


def _rodrigues_np(v):
    v = np.asarray(v, np.float64)
    t = np.linalg.norm(v + 1e-8)
    k = v / t
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(t) * K + (1 - math.cos(t)) * K @ K


def _joint_transforms_np(pose, transl, J, parents):
    n = len(parents)
    G = np.zeros((n, 4, 4))
    for i in range(n):
        L = np.eye(4)
        L[:3, :3] = _rodrigues_np(pose[3 * i:3 * i + 3])
        L[:3, 3] = J[i] - (J[parents[i]] if i > 0 else 0)
        G[i] = L if i == 0 else G[parents[i]] @ L
    A = G.copy()
    A[:, :3, 3] = G[:, :3, 3] - np.einsum("jab,jb->ja", G[:, :3, :3], J) + np.asarray(transl)[None]
    return A


def _bone_segments(J, parents):
    """Segment (start, end) bound to each joint: joint -> its first child (leaf: short stub)."""
    n = len(parents)
    children = [[] for _ in range(n)]
    for i in range(1, n):
        children[parents[i]].append(i)
    segs = []
    for j in range(n):
        if children[j]:
            segs.append([(J[j], J[c]) for c in children[j]])
        else:
            d = J[j] - J[parents[j]]
            segs.append([(J[j], J[j] + 0.5 * d)])
    return segs


def _dist_to_segments(P, segs):
    best = np.full(P.shape[0], np.inf)
    for a, b in segs:
        ab = b - a
        t = np.clip(((P - a) @ ab) / max(float(ab @ ab), 1e-12), 0, 1)
        d = np.linalg.norm(P - (a + t[:, None] * ab), axis=1)
        best = np.minimum(best, d)
    return best


def make_assets(num_points: int = 200_000, uv_size: int = 512, smpl_type: str = "smpl", seed: int = 0):
    """Returns a dict of CPU tensors / numpy arrays with the reference's shapes."""
    rng = np.random.default_rng(seed)
    if smpl_type == "smpl":
        parents, J = SMPL_PARENTS, _SMPL_REST.copy()
    else:
        parents, J = smplx_like_tree(seed)
    nj = len(parents)
    S = uv_size
    assert 0 < num_points <= S * S
    # ---- UV layout: one horizontal band of the S x S map per bone, height ~ bone surface
    segs = []
    for j in range(1, nj):
        a, b = J[parents[j]], J[j]
        r = _RADIUS.get(parents[j], 0.02) if nj == 24 or parents[j] < 22 else 0.012
        segs.append((a, b, r))
    segs.append((J[15], J[15] + np.array([0, 0.12, 0.0]), _RADIUS[15]))             # head
    area = np.array([max(np.linalg.norm(b - a), 0.03) * r for a, b, r in segs])
    rows = np.maximum(1, np.floor(area / area.sum() * S).astype(int))
    while rows.sum() > S:
        rows[np.argmax(rows)] -= 1
    while rows.sum() < S:
        rows[np.argmax(area / rows)] += 1
    band_of_row = np.repeat(np.arange(len(segs)), rows)
    row_in_band = np.concatenate([np.arange(r) for r in rows])
    rr, cc = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    bi = band_of_row[rr.reshape(-1)]
    tpar = (row_in_band[rr.reshape(-1)] + 0.5) / rows[bi]
    phi = (cc.reshape(-1) + 0.5) / S * 2 * math.pi
    A0 = np.stack([s[0] for s in segs])[bi]
    B0 = np.stack([s[1] for s in segs])[bi]
    R0 = np.array([s[2] for s in segs])[bi]
    axis = B0 - A0
    axis /= np.maximum(np.linalg.norm(axis, axis=1, keepdims=True), 1e-9)
    ref = np.where(np.abs(axis[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    e1 = np.cross(axis, ref)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(axis, e1)
    # torso cross-sections are wider in x than in z
    flat = np.where(np.isin(bi, [k for k, s in enumerate(segs) if s[2] >= 0.12]), 0.72, 1.0)
    rest_pts = (A0 + tpar[:, None] * (B0 - A0)
                + R0[:, None] * (np.cos(phi)[:, None] * e1 + (flat * np.sin(phi))[:, None] * e2))
    # ---- valid mask with exactly num_points texels
    valid = np.ones(S * S, dtype=bool)
    drop = rng.choice(S * S, S * S - num_points, replace=False)
    valid[drop] = False
    # ---- skinning weights from bone distances (rest pose), top-4, renormalised
    bones = _bone_segments(J, parents)
    P = rest_pts[valid]
    d = np.stack([_dist_to_segments(P, bones[j]) for j in range(nj)], 1)
    logit = -(d / 0.05) ** 2
    logit -= logit.max(1, keepdims=True)
    w = np.exp(logit)
    kth = np.partition(w, -4, axis=1)[:, -4][:, None]
    w[w < kth] = 0
    w /= w.sum(1, keepdims=True)
    # ---- canonical pose: legs +-30 degrees about z (arguments/__init__.py:44-53)
    cpose = np.zeros(nj * 3)
    cpose[5] = 30 / 180 * math.pi
    cpose[8] = -30 / 180 * math.pi
    A_cano = _joint_transforms_np(cpose, [0.0, 0.3, 0.0], J, parents)
    T = np.einsum("nj,jab->nab", w, A_cano)
    cano_pts = np.einsum("nab,nb->na", T[:, :3, :3], P) + T[:, :3, 3]
    # full S x S posmap (invalid texels keep their rest position; never read through valid_idx)
    query_posmap = rest_pts.copy()
    query_posmap[valid] = cano_pts
    lbs_map = np.zeros((S * S, nj), np.float32)
    lbs_map[valid] = w
    idx = np.stack([rr.reshape(-1), cc.reshape(-1)], 1).astype(np.float32) / (S - 1)
    tp = _load_test_pose()
    return dict(
        smpl_type=smpl_type, num_joints=nj, parents=np.asarray(parents, np.int32),
        joints_rest=torch.tensor(J, dtype=torch.float32),
        valid_idx=torch.tensor(valid), uv_coord_map=torch.tensor(idx),
        query_posmap=torch.tensor(query_posmap.reshape(S, S, 3), dtype=torch.float32),
        lbs_map=torch.tensor(lbs_map.reshape(S, S, nj)),
        cano_joint_mat=torch.tensor(A_cano, dtype=torch.float32)[None],
        betas=torch.tensor(tp["beta"], dtype=torch.float32),
        body_pose=torch.tensor(tp["body_pose"], dtype=torch.float32),
        trans=torch.tensor(tp["trans"], dtype=torch.float32),
        intrinsic=tp["intrinsic"].astype(np.float64), extrinsic=tp["extrinsic"].astype(np.float64))


def make_frames(assets: dict, num_frames: int, width: int, height: int, seed: int = 0):
    """Per-frame training data in the reference's dataset item format
    (/root/reference/scene/dataset_mono.py:176-257): pose/transl tables for the embeddings, one
    camera (the shipped test camera rescaled to width x height) and target images."""
    nj = assets["num_joints"]
    rng = np.random.default_rng(seed)
    bp = assets["body_pose"].numpy()
    tr = assets["trans"].numpy()
    sel = np.arange(num_frames) % bp.shape[0]
    pose72 = bp[sel]
    if nj == 24:
        pose = pose72
    else:   # SMPL-X layout: 22 body joints optimised (66), the rest comes from rest_pose
        pose = pose72[:, :66]
    transl = tr[sel] + rng.normal(0, 0.01, (num_frames, 3)).astype(np.float32)
    sx, sy = width / 1024.0, height / 1024.0
    s = min(sx, sy)
    K = assets["intrinsic"].copy()
    K[0, 0] *= s
    K[1, 1] *= s
    K[0, 2], K[1, 2] = width / 2.0, height / 2.0
    cam = make_camera(K, assets["extrinsic"], width, height)
    return dict(pose=torch.tensor(pose, dtype=torch.float32), transl=torch.tensor(transl, dtype=torch.float32),
                rest_pose=torch.zeros(num_frames, 99), camera=cam)


# ----------------------------------------------------------------------------- on-disk layout
def synthetic_body_model(assets: dict, seed: int = 0) -> dict:
    """An SMPL-file-shaped body model whose joint regression reproduces assets['joints_rest'] for
    every beta: four vertices around each joint, J_regressor = their mean, shape directions with
    zero mean per joint group. Keys as in the official files (v_template, shapedirs, J_regressor,
    kintree_table, weights, posedirs, f)."""
    rng = np.random.default_rng(seed)
    J = assets["joints_rest"].double().numpy()
    parents = np.asarray(assets["parents"], np.int64)
    nj = J.shape[0]
    off = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]], np.float64) * 0.03
    v_template = (J[:, None, :] + off[None]).reshape(nj * 4, 3)
    J_regressor = np.zeros((nj, nj * 4))
    for j in range(nj):
        J_regressor[j, 4 * j:4 * j + 4] = 0.25
    sd = rng.normal(0, 0.01, (nj, 4, 3, 10))
    sd -= sd.mean(axis=1, keepdims=True)
    kintree = np.stack([parents, np.arange(nj)]).astype(np.int64)
    kintree[0, 0] = 2 ** 32 - 1                     # the official files store -1 as uint32
    weights = np.repeat(np.eye(nj), 4, axis=0)
    faces = np.stack([np.arange(0, nj * 4 - 2), np.arange(1, nj * 4 - 1), np.arange(2, nj * 4)], 1)
    return dict(v_template=v_template, shapedirs=sd.reshape(nj * 4, 3, 10), J_regressor=J_regressor,
                kintree_table=kintree.astype(np.uint32), weights=weights,
                posedirs=np.zeros((nj * 4, 3, (nj - 1) * 9)), f=faces.astype(np.uint32))


def write_dataset(source_path: str, project_path: str, assets: dict, frames: dict, images=None, masks=None,
                  splits=("train", "test"), inp_posmap_size: int = 128, stage2: bool = True, seed: int = 0):
    """Writes the synthetic assets in the reference's on-disk layout (see dataset.py's header for
    the file list), so that AvatarModel(source_path=...) / MonoDataset_* read them back.

    images [F,3,H,W] in [0,1] (default: white), masks [F,H,W] bool (default: all foreground).
    Returns a dict of the paths a ModelParams needs (source_path, project_path, smpl_model_path,
    smplx_model_path, test_folder)."""
    import pickle

    from PIL import Image
    import scipy.sparse
    st = assets["smpl_type"]
    S = assets["query_posmap"].shape[0]
    nj = assets["num_joints"]
    cam = frames["camera"]
    W, H = cam["width"], cam["height"]
    F = frames["pose"].shape[0]
    rng = np.random.default_rng(seed)
    # ---- project assets
    os.makedirs(os.path.join(project_path, "assets", "uv_masks"), exist_ok=True)
    valid = assets["valid_idx"].numpy().reshape(-1)
    faceid = np.full(S * S, -1, np.int64)
    faceid[valid] = np.arange(valid.sum()) % (nj * 4 - 2)
    np.save(os.path.join(project_path, "assets", "uv_masks", "uv_mask{}_with_faceid_{}.npy".format(S, st)),
            faceid.reshape(S, S))
    np.save(os.path.join(project_path, "assets", "lbs_map_{}_{}.npy".format(st, S)), assets["lbs_map"].numpy())
    body = synthetic_body_model(assets, seed)
    np.save(os.path.join(project_path, "assets", "{}_faces.npy".format(st)), body["f"])
    model_dir = os.path.join(project_path, "assets", "smpl_files", st)
    os.makedirs(model_dir, exist_ok=True)
    if st == "smpl":       # official SMPL: latin1 pickle with a scipy.sparse joint regressor
        blob = dict(body, J_regressor=scipy.sparse.csc_matrix(body["J_regressor"]))
        with open(os.path.join(model_dir, "SMPL_NEUTRAL.pkl"), "wb") as f:
            pickle.dump(blob, f, protocol=2)
    else:
        np.savez(os.path.join(model_dir, "SMPLX_NEUTRAL.npz"), **body)
    def write_inp_map(folder, i):
        # stand-in for the posed-body position maps of scripts/gen_pose_map_our_smpl.py
        pm = (rng.standard_normal((inp_posmap_size, inp_posmap_size, 3)) * 0.3).astype(np.float32)
        np.savez(os.path.join(folder, "inp_map", "inp_posemap_%s_%s.npz" % (inp_posmap_size, "%08d" % i)),
                 **{"posmap" + str(inp_posmap_size): pm})

    # ---- per-split data
    body_pose = frames["pose"] if st == "smpl" else torch.cat([frames["pose"], frames["rest_pose"]], dim=1)
    parms = {"beta": assets["betas"][:1].clone(), "trans": frames["transl"].clone(), "body_pose": body_pose.clone()}
    for split in splits:
        folder = os.path.join(source_path, split)
        for sub in ("images", "masks", "cam_parms", "inp_map"):
            os.makedirs(os.path.join(folder, sub), exist_ok=True)
        torch.save(parms, os.path.join(folder, "smpl_parms.pth"))
        torch.save(parms, os.path.join(folder, "smpl_parms_pred.pth"))
        np.savez(os.path.join(folder, "cam_parms.npz"), extrinsic=cam["extrinsic"], intrinsic=cam["intrinsic"])
        np.savez(os.path.join(folder, "query_posemap_{}_cano_{}.npz".format(S, st)),
                 **{"posmap" + str(S): assets["query_posmap"].numpy()})
        if inp_posmap_size != S:
            step = S // inp_posmap_size
            np.savez(os.path.join(folder, "query_posemap_{}_cano_{}.npz".format(inp_posmap_size, st)),
                     **{"posmap" + str(inp_posmap_size): assets["query_posmap"].numpy()[::step, ::step]})
        torch.save(assets["cano_joint_mat"][0].clone(), os.path.join(folder, "{}_cano_joint_mat.pth".format(st)))
        for i in range(F):
            name = "%08d" % i
            img = np.full((H, W, 3), 255, np.uint8) if images is None else \
                (images[i].permute(1, 2, 0).clamp(0, 1) * 255 + 0.5).to(torch.uint8).numpy()
            Image.fromarray(img, "RGB").save(os.path.join(folder, "images", name + ".png"))
            m = np.full((H, W), 255, np.uint8) if masks is None else (masks[i].numpy().astype(np.uint8) * 255)
            Image.fromarray(m, "L").save(os.path.join(folder, "masks", name + ".png"))
            np.savez(os.path.join(folder, "cam_parms", name + ".npz"), extrinsic=cam["extrinsic"], intrinsic=cam["intrinsic"])
            if stage2:
                write_inp_map(folder, i)
    # ---- novel-pose folder (assets/test_pose of the reference: pose table + static camera)
    test_folder = os.path.join(project_path, "assets", "test_pose")
    os.makedirs(os.path.join(test_folder, "inp_map"), exist_ok=True)
    if stage2:
        for i in range(F):
            write_inp_map(test_folder, i)
    torch.save(parms, os.path.join(test_folder, "smpl_parms.pth"))
    np.savez(os.path.join(test_folder, "cam_parms.npz"), extrinsic=cam["extrinsic"], intrinsic=cam["intrinsic"])
    return dict(source_path=source_path, project_path=project_path,
                smpl_model_path=os.path.join(project_path, "assets", "smpl_files", "smpl"),
                smplx_model_path=os.path.join(project_path, "assets", "smpl_files", "smplx"),
                test_folder=test_folder)
