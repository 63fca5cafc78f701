"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON CODE in the build
container (python oracle/make_golden.py). /root/reference does not exist on the GPU box, so
the vectors are committed; this script is the provenance record.

What is imported from /root/reference (read-only), unmodified:
    submodules.smplx.lbs        lbs(), batch_rodrigues(), batch_rigid_transform()
    model.network               POP_no_unet
    model.modules               UnetNoCond5DS, ShapeDecoder, GeomConvLayers
    utils.graphics_utils        geom_transform_points, getWorld2View2, getProjectionMatrix, focal2fov
    utils.loss_utils            l1_loss_w, ssim
    utils.general_utils         getIdxMap_torch
Two lines that cannot be imported (model/avatar_model.py needs trimesh + CUDA at import) are
restated verbatim in spirit: the skinning einsums of model/avatar_model.py:311-314.
The rasterizer has no reference implementation in the tree (parity unpinned, see
oracle/gsr_oracle.c); its golden vectors (raster_golden.npz) are produced by the C oracle and
serve as regression vectors, not as pins.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]


def synthetic_body(V, Jn, parents, seed):
    g = torch.Generator().manual_seed(seed)
    v_template = torch.randn(V, 3, generator=g) * torch.tensor([0.3, 0.8, 0.15])
    shapedirs = torch.randn(V, 3, 10, generator=g) * 0.01
    posedirs = torch.randn((Jn - 1) * 9, V * 3, generator=g) * 0.001
    Jr = torch.rand(Jn, V, generator=g)
    Jr = Jr / Jr.sum(1, keepdim=True)
    w = torch.rand(V, Jn, generator=g) ** 8
    w = w / w.sum(1, keepdim=True)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=Jr,
                parents=torch.tensor(parents, dtype=torch.long), lbs_weights=w)


def make_lbs():
    from submodules.smplx.lbs import lbs, batch_rodrigues, batch_rigid_transform
    parms = torch.load(os.path.join(REF, "assets/test_pose/smpl_parms.pth"))
    out = {}
    # SMPL-shaped (24 joints, the real kinematic tree) with poses shipped by the reference
    body = synthetic_body(64, 24, SMPL_PARENTS, seed=0)
    rows = [0, 37, 111, 250, 479]
    pose = torch.cat([torch.zeros(1, 72), parms["body_pose"][rows]], 0)          # incl. T-pose
    transl = torch.cat([torch.zeros(1, 3), parms["trans"][rows]], 0)
    betas = parms["beta"].expand(pose.shape[0], -1).contiguous()
    verts, joints, A = lbs(betas, pose, body["v_template"], body["shapedirs"], body["posedirs"],
                           body["J_regressor"], body["parents"], body["lbs_weights"],
                           pose2rot=True, return_affine_mat=True)
    A = A.clone()
    A[:, :, :3, 3] += transl.unsqueeze(dim=1)            # body_models.py:383
    R = batch_rodrigues(pose.view(-1, 3))
    out.update(smpl_pose=pose, smpl_transl=transl, smpl_betas=betas, smpl_A=A, smpl_rodrigues=R, smpl_verts=verts,
               smpl_joints=joints,
               **{"smpl_" + k: v for k, v in body.items()})
    # SMPL-X-shaped: 55 joints, random valid tree, random poses (config 5)
    g = torch.Generator().manual_seed(1)
    parents55 = [-1] + [int(torch.randint(0, i, (1,), generator=g)) for i in range(1, 55)]
    body55 = synthetic_body(96, 55, parents55, seed=2)
    pose55 = torch.randn(3, 165, generator=g) * 0.4
    transl55 = torch.randn(3, 3, generator=g)
    betas55 = torch.randn(1, 10, generator=g).expand(3, -1).contiguous()
    verts55, joints55, A55 = lbs(betas55, pose55, body55["v_template"], body55["shapedirs"], body55["posedirs"],
                    body55["J_regressor"], body55["parents"], body55["lbs_weights"],
                    pose2rot=True, return_affine_mat=True)
    A55 = A55.clone()
    A55[:, :, :3, 3] += transl55.unsqueeze(dim=1)
    out.update(smplx_pose=pose55, smplx_transl=transl55, smplx_betas=betas55, smplx_A=A55, smplx_verts=verts55,
               smplx_joints=joints55,
               **{"smplx_" + k: v for k, v in body55.items()})
    # chain only (batch_rigid_transform), arbitrary rotations
    Rm = batch_rodrigues(torch.randn(2 * 24, 3, generator=g)).view(2, 24, 3, 3)
    Jrest = torch.randn(2, 24, 3, generator=g)
    posed, rel = batch_rigid_transform(Rm, Jrest, torch.tensor(SMPL_PARENTS))
    out.update(chain_R=Rm, chain_J=Jrest, chain_posed=posed, chain_A=rel)
    np.savez_compressed(os.path.join(OUT, "lbs_golden.npz"), **{k: v.numpy() for k, v in out.items()})
    return out


def make_skin(lbs_out):
    g = torch.Generator().manual_seed(3)
    B, N, Jn = 2, 700, 24
    A = lbs_out["smpl_A"][1:3]
    inv_mats = torch.linalg.inv(lbs_out["smpl_A"][0:1]).expand(B, -1, -1, -1)
    cano2live = torch.matmul(A, inv_mats)                                          # avatar_model.py:296
    query_points = (torch.randn(1, N, 3, generator=g) * 0.4).expand(B, -1, -1)
    res = torch.randn(B, N, 3, generator=g) * 0.02
    w = torch.rand(N, Jn, generator=g) ** 6
    w[w < 0.05] = 0
    w = (w / w.sum(1, keepdim=True).clamp(min=1e-8))[None].expand(B, -1, -1).contiguous()
    cano_deform_point = res + query_points                                         # :309
    pt_mats = torch.einsum('bnj,bjxy->bnxy', w, cano2live)                         # :311
    full_pred = torch.einsum('bnxy,bny->bnx', pt_mats[..., :3, :3], cano_deform_point) + pt_mats[..., :3, 3]
    np.savez_compressed(os.path.join(OUT, "skin_golden.npz"), A=A.numpy(), inv_mats=inv_mats.numpy(),
                        cano2live=cano2live.numpy(), query_points=query_points.numpy(),
                        res=res.numpy(), weights=w.numpy(), full_pred=full_pred.numpy())


def make_net():
    from model.network import POP_no_unet
    from model.modules import UnetNoCond5DS
    from utils.general_utils import getIdxMap_torch
    torch.manual_seed(4)
    net = POP_no_unet(c_geom=8, geom_layer_type='conv', nf=4, hsize=16, up_mode='upconv',
                      use_dropout=False, uv_feat_dim=2)
    net.train()
    B, S_in, S_q = 2, 16, 32
    geom = torch.randn(B, 8, S_in, S_in) * 0.5
    posef = torch.randn(B, 8, S_in, S_in) * 0.5
    uv = getIdxMap_torch(torch.rand(3, S_q, S_q))[None].expand(B, -1, -1).contiguous()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    r1, s1, c1 = net(None, geom, uv)                 # stage-1 call (pose_featmap=None)
    net.load_state_dict(sd0)                         # reset BN running stats
    r2, s2, c2 = net(posef, geom, uv)                # stage-2 call
    save = {"net." + k: v.numpy() for k, v in sd0.items()}
    save.update(geom=geom.numpy(), posef=posef.numpy(), uv=uv.numpy(),
                res1=r1.detach().numpy(), scales1=s1.detach().numpy(), shs1=c1.detach().numpy(),
                res2=r2.detach().numpy(), scales2=s2.detach().numpy(), shs2=c2.detach().numpy())
    unet = UnetNoCond5DS(input_nc=3, output_nc=8, nf=4, up_mode='upconv', use_dropout=False)
    unet.train()
    x = torch.randn(2, 3, 64, 64)
    usd = {k: v.clone() for k, v in unet.state_dict().items()}
    y = unet(x.clone())
    save.update({"unet." + k: v.numpy() for k, v in usd.items()})
    save.update(unet_x=x.numpy(), unet_y=y.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "net_golden.npz"), **save)


def fill_state(module, seed):
    """Deterministic parameters that do not depend on any module's init order: every float entry
    of the state dict, in sorted key order, is drawn from one seeded generator (weights
    N(0, 1/fan_in)-ish, BatchNorm weight around 1, running_var positive). tests/ uses the same
    function (imported from here) to rebuild the state the golden outputs belong to."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for k in sorted(sd):
        v = sd[k]
        if not v.is_floating_point():
            continue
        r = torch.randn(v.shape, generator=g)
        if k.endswith("running_var"):
            v.copy_(0.5 + r.abs())
        elif v.dim() == 1 and k.endswith("weight"):          # norm scale
            v.copy_(1.0 + 0.2 * r)
        elif v.dim() == 1:                                     # biases, norm shift, running_mean
            v.copy_(0.1 * r)
        else:
            fan_in = v[0].numel()
            v.copy_(r * (1.5 / fan_in ** 0.5))
    module.load_state_dict(sd)
    return module


def net_full_inputs():
    """Seeded inputs of make_net_full (regenerated by the tests instead of being stored)."""
    B, S_in, S_q = 2, 32, 64
    g = torch.Generator().manual_seed(22)
    geom = torch.randn(1, 64, S_in, S_in, generator=g) * 0.5
    posef = torch.randn(B, 64, S_in, S_in, generator=g) * 0.5
    w = [torch.randn(B, c, S_q * S_q, generator=g) for c in (3, 1, 3)]
    x = torch.randn(2, 3, 64, 64, generator=g) * 0.3
    wy = torch.randn(2, 8, 64, 64, generator=g)
    r, c = torch.meshgrid(torch.arange(S_q), torch.arange(S_q), indexing="ij")
    uv = (torch.stack([r.reshape(-1), c.reshape(-1)], 1).float() / (S_q - 1))[None].expand(B, -1, -1).contiguous()
    return dict(B=B, S_in=S_in, S_q=S_q, geom=geom, posef=posef, w=w, unet_x=x, unet_wy=wy, uv=uv)


NET_FULL_KEEP = ["decoder.conv1.weight", "decoder.conv5.weight", "decoder.conv4.bias", "decoder.bn3.weight",
                 "decoder.bn7.bias", "decoder.conv8.weight", "decoder.conv8SH.weight", "decoder.conv8N.weight",
                 "decoder.conv7SH.weight", "geom_proc_layers.conv1.weight", "geom_proc_layers.conv3.weight"]


def make_net_full():
    """The reference's POP_no_unet / UnetNoCond5DS at the PRODUCTION widths
    (arguments/__init__.py:101-111: c_geom 64, c_pose 64, hsize 128, nf 32) — the widths that select the
    fused MFMA decoder and the fused up-sampling here. Outputs and gradients (the large ones subsampled:
    every 2nd / 4th channel, first 8 output channels of the 5x5 conv weights); weights by fill_state,
    inputs by net_full_inputs."""
    from model.network import POP_no_unet
    from model.modules import UnetNoCond5DS
    from utils.general_utils import getIdxMap_torch
    net = fill_state(POP_no_unet(c_geom=64, geom_layer_type='conv', nf=32, hsize=128, up_mode='upconv',
                                 use_dropout=False, uv_feat_dim=2), seed=21)
    net.train()
    inp = net_full_inputs()
    B, S_q = inp["B"], inp["S_q"]
    geom = inp["geom"].requires_grad_(True)
    posef = inp["posef"].requires_grad_(True)
    uv = getIdxMap_torch(torch.rand(3, S_q, S_q))[None].expand(B, -1, -1).contiguous()
    assert torch.equal(uv, inp["uv"])
    w = inp["w"]
    save = {}
    for tag, pf in (("s1", None), ("s2", posef)):
        net.zero_grad()
        geom.grad = None
        posef.grad = None
        outs = net(pf, geom.expand(B, -1, -1, -1).contiguous(), uv)           # avatar_model.py:298-306
        loss = sum((o * wi).sum() for o, wi in zip(outs, w))
        loss.backward()
        for name, o in zip(("res", "scales", "shs"), outs):
            save[f"{tag}_{name}"] = o.detach().numpy()
        save[f"{tag}_dgeom"] = geom.grad[:, ::2].numpy().copy()
        if pf is not None:
            save[f"{tag}_dposef"] = posef.grad[:, ::4].numpy().copy()
        params = dict(net.named_parameters())
        for k in NET_FULL_KEEP:
            gk = params[k].grad
            save[f"{tag}_d.{k}"] = (gk[:8] if k.startswith("geom_proc") else gk).numpy().copy()
    unet = fill_state(UnetNoCond5DS(input_nc=3, output_nc=64, nf=32, up_mode='upconv', use_dropout=False), seed=23)
    unet.train()
    y = unet(inp["unet_x"].clone())
    (y[:, ::8] * inp["unet_wy"]).sum().backward()
    up = dict(unet.named_parameters())
    save.update(unet_y=y[:, ::8].detach().numpy())
    for k in sorted(up):
        if up[k].numel() <= 20000:
            save["unet_d." + k] = up[k].grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "net_full_golden.npz"), **save)
    print("net_full_golden.npz", os.path.getsize(os.path.join(OUT, "net_full_golden.npz")), len(save), "arrays")


def make_camera_loss():
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix, focal2fov, geom_transform_points
    from utils.loss_utils import l1_loss_w, ssim
    c = np.load(os.path.join(REF, "assets/test_pose/cam_parms.npz"))
    K = c["intrinsic"].astype(np.float64)
    E = c["extrinsic"]
    out = {}
    for size in (1024, 512, 256):
        s = size / 1024.0
        Ks = K.copy()
        Ks[:2] *= s
        fovx, fovy = focal2fov(Ks[0, 0], size), focal2fov(Ks[1, 1], size)
        R, T = np.transpose(E[:3, :3]), E[:3, 3]
        wvt = torch.tensor(getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        # python floats for K: the reference's float32-numpy K breaks under numpy>=2
        Kn = np.array([[float(v) for v in row] for row in Ks])
        proj = getProjectionMatrix(znear=0.01, zfar=100, fovX=fovx, fovY=fovy, K=Kn, h=size, w=size).transpose(0, 1)
        full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
        out[f"wvt_{size}"] = wvt.numpy()
        out[f"full_{size}"] = full.numpy()
        out[f"center_{size}"] = wvt.inverse()[3, :3].numpy()
        out[f"fov_{size}"] = np.array([fovx, fovy])
    g = torch.Generator().manual_seed(5)
    pts = torch.randn(50, 3, generator=g) * 0.5
    out["proj_pts"] = pts.numpy()
    out["proj_out"] = geom_transform_points(pts, torch.tensor(out["full_1024"])).numpy()
    a = torch.rand(2, 3, 40, 56, generator=g)
    b = (a + 0.1 * torch.randn(2, 3, 40, 56, generator=g)).clamp(0, 1)
    out.update(loss_a=a.numpy(), loss_b=b.numpy(), l1=np.array(l1_loss_w(a, b).item()),
               ssim=np.array(ssim(a, b).item()))
    np.savez_compressed(os.path.join(OUT, "camera_loss_golden.npz"), **out)
    # the poses the reference ships (first 16 of 480) + the camera, for bench/synthetic data
    parms = torch.load(os.path.join(REF, "assets/test_pose/smpl_parms.pth"))
    np.savez_compressed(os.path.join(OUT, "test_pose.npz"), intrinsic=c["intrinsic"], extrinsic=c["extrinsic"],
                        beta=parms["beta"].numpy(), body_pose=parms["body_pose"][::30].numpy(),
                        trans=parms["trans"][::30].numpy())


def make_raster():
    """Regression vectors from the C oracle (NOT a pin — see module docstring)."""
    from oracle.gsr_oracle import RasterOracle
    from tests.scenes import cam_kwargs, random_scene
    O = RasterOracle()
    save = {}
    for name, kind in (("gen", "general"), ("ava", "avatar")):
        sc = random_scene(64, 48, 32, seed=123, kind=kind, scale_med=0.06)
        st = O.forward(sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"], **cam_kwargs(sc))
        g = np.random.default_rng(9).normal(0, 1, (3, 32, 48)).astype(np.float32)
        b = O.backward(st, g)
        for k in ("means3D", "colors", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "bg"):
            save[f"{name}_{k}"] = sc[k]
        save[f"{name}_tan"] = np.array([sc["tanfovx"], sc["tanfovy"]])
        for k in ("color", "radii", "rect", "tiles_touched", "ranges", "point_list", "n_contrib", "final_T"):
            save[f"{name}_{k}"] = st[k]
        save[f"{name}_g"] = g
        for k in ("dmeans3D", "dcolors", "dopacity", "dscales", "drots", "dmeans2D"):
            save[f"{name}_{k}"] = b[k]
    np.savez_compressed(os.path.join(OUT, "raster_golden.npz"), **save)


def make_dataset():
    """Items of the reference's MonoDataset_train/_test/_novel_pose/_novel_view
    (scene/dataset_mono.py) and its to_cuda / getIdxMap_torch on a dataset written by
    synthetic.write_dataset. scene.dataset_mono imports cv2 (absent here) for one Rodrigues
    call: a numpy stand-in module is registered for the import."""
    import tempfile
    import types
    from types import SimpleNamespace
    from tests.scenes import dataset_fixture

    def rodrigues(v):
        v = np.asarray(v, np.float64).reshape(3)
        th = np.linalg.norm(v)
        if th < 1e-12:
            return np.eye(3), None
        k = v / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K), None
    sys.modules.setdefault("cv2", types.SimpleNamespace(Rodrigues=rodrigues))
    from scene import dataset_mono as R
    # dataset_mono.py:222 hands an int8 array to Image.fromarray(..., "RGB"); the Pillow the
    # reference pins reinterprets the bytes, Pillow 12 here refuses the dtype. Same bytes:
    _fromarray = R.Image.fromarray
    R.Image.fromarray = lambda a, mode=None: _fromarray(a.view(np.uint8) if a.dtype == np.int8 else a, mode)
    # torch 2.10 refuses `tensor[i, j] = numpy.float32` (graphics_utils.py:65 with the float32 K of
    # dataset_mono.py:166): hand K over as float64 (exact upcast of the same values).
    _proj = R.getProjectionMatrix
    R.getProjectionMatrix = lambda **kw: _proj(**dict(kw, K=np.asarray(kw["K"], np.float64)))
    from utils.general_utils import to_cuda, getIdxMap_torch
    out = {}
    for st in ("smpl", "smplx"):
        with tempfile.TemporaryDirectory() as tmp:
            assets, frames, paths = dataset_fixture(tmp, st)
            for stage in (1, 2):
                for cam_static in (1, 0):
                    parms = SimpleNamespace(train_stage=stage, smpl_type=st, smpl_gender="neutral", no_mask=0,
                                            cam_static=cam_static, inp_posmap_size=16, query_posmap_size=32, **paths)
                    sets = {"train": R.MonoDataset_train(parms), "test": R.MonoDataset_test(parms)}
                    if cam_static:
                        sets["novel_pose"] = R.MonoDataset_novel_pose(parms)
                        nv = R.MonoDataset_novel_view(parms)
                        # update_smpl() needs the licensed numpy SMPL model; set what it computes
                        nv.Th = assets["joints_rest"][0].double().numpy() + nv.smpl_data["trans"][2].double().numpy()
                        nv.data_length, nv.fix_pose_idx = 5, 2
                        sets["novel_view"] = nv
                    for name, ds in sets.items():
                        for i in (0, 3):
                            item = ds[i]
                            batch = to_cuda({k: (v if not isinstance(v, (int, float)) else torch.tensor([v]))
                                             for k, v in item.items()}, device="cpu")
                            for k, v in batch.items():
                                out["%s/s%d/c%d/%s/%d/%s" % (st, stage, cam_static, name, i, k)] = np.asarray(v)
                        out["%s/s%d/c%d/%s/len" % (st, stage, cam_static, name)] = np.asarray(len(ds))
    out["idx_map_8"] = getIdxMap_torch(torch.rand(3, 8, 8)).numpy()
    np.savez_compressed(os.path.join(OUT, "dataset_golden.npz"), **out)
    print("dataset_golden.npz", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["dataset"]:
        make_dataset()
        sys.exit(0)
    if sys.argv[1:] == ["net_full"]:
        make_net_full()
        sys.exit(0)
    if sys.argv[1:] == ["lbs"]:
        make_lbs()
        sys.exit(0)
    lo = make_lbs()
    make_skin(lo)
    make_net()
    make_net_full()
    make_camera_loss()
    make_raster()
    make_dataset()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
