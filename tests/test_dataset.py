"""CPU tests of the data layer (gaussianavatar_amd/dataset.py): our readers of the reference's
on-disk formats against items produced by the reference's own MonoDataset_* classes on the same
files (tests/golden/dataset_golden.npz, made by oracle/make_golden.py), the asset round trip, and
the model built from disk against the model built from the in-memory generator."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from gaussianavatar_amd import dataset as D
from gaussianavatar_amd.avatar_model import AvatarModel, default_params
from tests.scenes import dataset_fixture

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataset_golden.npz")


def _parms(paths, st, stage, cam_static=1):
    return SimpleNamespace(train_stage=stage, smpl_type=st, smpl_gender="neutral", no_mask=0, cam_static=cam_static,
                           inp_posmap_size=16, query_posmap_size=32, **paths)


@pytest.mark.parametrize("st", ["smpl", "smplx"])
def test_dataset_items_match_reference_golden(tmp_path, st):
    gold = np.load(GOLD)
    assets, _frames, paths = dataset_fixture(str(tmp_path), st)
    checked = 0
    for stage in (1, 2):
        for cam_static in (1, 0):
            p = _parms(paths, st, stage, cam_static)
            sets = {"train": D.MonoDataset_train(p), "test": D.MonoDataset_test(p)}
            if cam_static:
                sets["novel_pose"] = D.MonoDataset_novel_pose(p)
                nv = D.MonoDataset_novel_view(p, joints_rest=assets["joints_rest"])
                nv.update_smpl(2, 5)
                sets["novel_view"] = nv
            for name, ds in sets.items():
                prefix = "%s/s%d/c%d/%s/" % (st, stage, cam_static, name)
                assert len(ds) == int(gold[prefix + "len"])
                for i in (0, 3):
                    item = ds[i]
                    keys = {k[len(prefix) + 2:] for k in gold.files if k.startswith(prefix + "%d/" % i)}
                    assert set(item.keys()) == keys, (prefix, set(item.keys()) ^ keys)
                    for k in keys:
                        want = gold[prefix + "%d/%s" % (i, k)]
                        got = np.asarray(item[k], dtype=want.dtype).reshape(want.shape)
                        if want.dtype.kind in "iub":
                            assert (got == want).all(), (prefix, i, k)
                        else:       # float32 camera algebra (two 4x4 inversions) / 8-bit images
                            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6, err_msg=prefix + k)
                        checked += 1
    assert checked > 250


def test_uv_index_map_matches_reference_golden():
    np.testing.assert_array_equal(D.uv_index_map(8).numpy(), np.load(GOLD)["idx_map_8"])


def test_to_cuda_semantics():
    items = {"a": np.ones((2, 2), np.float64), "b": torch.arange(3), "c": 1.5, "d": {"x": np.zeros(2)}}
    out = D.to_cuda(items, "cpu")
    assert out["a"].dtype == torch.float32 and out["b"].dtype == torch.int64 and out["c"] == 1.5
    assert out["d"]["x"].dtype == torch.float32
    out = D.to_cuda(items, "cpu", add_batch=True)
    assert out["a"].shape == (1, 2, 2) and out["c"] == [1.5] and out["d"]["x"].shape == (1, 2)
    with pytest.raises(TypeError):
        D.to_cuda({"d": {"x": 3}}, "cpu")


@pytest.mark.parametrize("st", ["smpl", "smplx"])
def test_assets_round_trip_through_disk(tmp_path, st):
    assets, frames, paths = dataset_fixture(str(tmp_path), st)
    mp, _n, _o = default_params(smpl_type=st, query_posmap_size=32, inp_posmap_size=16, **paths)
    back = D.load_assets(mp, "train")
    for k in ("valid_idx", "uv_coord_map", "query_posmap", "lbs_map", "cano_joint_mat"):
        assert torch.equal(back[k].float(), assets[k].float()), k
    # body model file (pickle with scipy.sparse regressor / npz) -> J(betas) for the dataset's betas
    assert (back["joints_rest"] - assets["joints_rest"]).abs().max() < 1e-6
    assert (back["parents"] == assets["parents"]).all()
    flist, valid, _uv = D.load_masks(paths["project_path"], 32, st)
    assert flist.shape == (int(valid.sum()), 3)


def test_body_model_pickle_with_chumpy_objects(tmp_path):
    """The official SMPL pickles hold chumpy.Ch arrays; they load without chumpy installed."""
    import pickle
    import sys
    import types
    ch = types.ModuleType("chumpy")
    chch = types.ModuleType("chumpy.ch")

    class Ch:
        def __init__(self, x):
            self.x = np.asarray(x)
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    chch.Ch = Ch
    ch.ch = chch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = ch, chch
    try:
        rng = np.random.default_rng(0)
        blob = dict(v_template=Ch(rng.normal(size=(12, 3))), shapedirs=Ch(rng.normal(size=(12, 3, 10))),
                    J_regressor=rng.random((3, 12)), kintree_table=np.array([[2 ** 32 - 1, 0, 1], [0, 1, 2]], np.uint32))
        with open(tmp_path / "SMPL_NEUTRAL.pkl", "wb") as f:
            pickle.dump(blob, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    body = D.load_body_model(str(tmp_path), "smpl", "neutral")
    assert body["parents"].tolist() == [-1, 0, 1]
    np.testing.assert_allclose(body["v_template"].numpy(), blob["v_template"].x.astype(np.float32))
    J = D.rest_joints(body, np.zeros(10))
    np.testing.assert_allclose(J.numpy(), (blob["J_regressor"] @ blob["v_template"].x).astype(np.float32), rtol=1e-5)


@pytest.mark.parametrize("stage", [1, 2])
def test_model_from_disk_equals_model_from_memory(tmp_path, stage):
    assets, frames, paths = dataset_fixture(str(tmp_path), "smpl")
    kw = dict(train_stage=stage, query_posmap_size=32, inp_posmap_size=16)
    torch.manual_seed(0)
    m_disk = AvatarModel(*default_params(**kw, **paths), device="cpu")
    torch.manual_seed(0)
    m_mem = AvatarModel(*default_params(**kw), assets=assets, frames=frames, device="cpu")
    assert m_disk.from_disk and not m_mem.from_disk
    for name in ("query_points", "query_lbs", "inv_mats", "betas", "valid_index", "uv_coord_map"):
        a, b = getattr(m_disk, name), getattr(m_mem, name)
        assert a.shape == b.shape and (a.float() - b.float()).abs().max() < 1e-6, name
    assert torch.equal(m_disk.pose.weight, m_mem.pose.weight) and torch.equal(m_disk.transl.weight, m_mem.transl.weight)
    assert (m_disk.smpl_model.joints_rest - m_mem.smpl_model.joints_rest).abs().max() < 1e-6
    loader = m_disk.getTrainDataloader()
    assert len(loader) == 2
    batch = next(iter(loader))
    assert batch["original_image"].shape == (2, 3, 48, 64) and batch["pose_idx"].dtype == torch.long
    assert isinstance(batch["FovX"], list) and isinstance(batch["width"][0], int)
    if stage == 2:
        assert batch["inp_pos_map"].shape == (2, 3, 16, 16) and batch["inp_pos_map"].dtype == torch.float32
    for getter in (m_disk.getTestDataset, m_disk.getNovelposeDataset):
        assert "pose_data" in getter()[0]


def test_frame_cache_and_uint8_transfer_path(tmp_path):
    """The training loader's fast path: frames leave the reader as the composited uint8 [3,H,W] and become the
    reference's float image on the consumer's side (/ 255, avatar_model._DeviceLoader) — bit-identical to the float
    item; decoded frames are cached per process within a byte budget (least recently used out)."""
    _assets, _frames, paths = dataset_fixture(str(tmp_path), "smpl")
    p = _parms(paths, "smpl", 1)
    ds = D.MonoDataset_train(p)
    ref = [ds[i]["original_image"] for i in range(len(ds))]
    assert ref[0].dtype == torch.float32 and len(ds._frames) == len(ds)          # all four frames cached
    again = ds[1]["original_image"]
    assert torch.equal(again, ref[1])
    ds.raw_uint8 = True
    for i in range(len(ds)):
        u = ds[i]["original_image"]
        assert u.dtype == torch.uint8 and u.shape == ref[i].shape and u.is_contiguous()
        assert torch.equal(u.float().div_(255.0), ref[i])
    # a budget of one frame: the cache never holds more
    p2 = _parms(paths, "smpl", 1)
    p2.cache_mb = 0
    ds2 = D.MonoDataset_train(p2)
    for i in range(len(ds2)):
        assert torch.equal(ds2[i]["original_image"], ref[i])
    assert len(ds2.__dict__.get("_frames", {})) == 0


def test_training_loader_serves_later_epochs_from_the_resident_cache(tmp_path):
    """The first epoch streams through the DataLoader and leaves every sample on the device; the following epochs
    are assembled from that cache with the DataLoader's own batch sampler. Every batch of every epoch must hold
    exactly the dataset's items for its pose_idx (image = float / 255 of the composited frame); a budget the
    dataset does not fit keeps streaming."""
    _assets, _frames, paths = dataset_fixture(str(tmp_path), "smpl")
    m = AvatarModel(*default_params(train_stage=1, query_posmap_size=32, inp_posmap_size=16, **paths), device="cpu")
    plain = D.MonoDataset_train(m.model_parms)                 # float items, the reference's format
    loader = m.getTrainDataloader()
    seen = []
    for epoch in range(3):
        n = 0
        for batch in loader:
            assert batch["original_image"].dtype == torch.float32 and batch["original_image"].shape == (2, 3, 48, 64)
            for b, i in enumerate(batch["pose_idx"].tolist()):
                ref = plain[i]
                assert torch.equal(batch["original_image"][b], ref["original_image"])
                assert torch.equal(batch["full_proj_transform"][b], ref["full_proj_transform"])
                assert batch["FovX"][b] == ref["FovX"] and batch["width"][b] == ref["width"]
                seen.append(i)
            n += 1
        assert n == len(loader) == 2
        assert len(loader.samples) == 4 and not loader.streaming   # all four frames resident after the first epoch
    assert sorted(seen) == sorted(list(range(4)) * 3)
    from gaussianavatar_amd.avatar_model import _DeviceLoader
    small = _DeviceLoader(loader.loader, "cpu", budget_gb=1e-5)
    for _ in range(2):
        assert sum(1 for _b in small) == 2
    assert small.streaming and not small.samples
