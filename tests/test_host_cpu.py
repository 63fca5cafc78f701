"""CPU tests of host-side pieces added in round 5: the GA_DEV switches, AvatarModel.close(), the training loader's budget."""
import gc

import pytest
import torch

from gaussianavatar_amd import _dev


def test_ga_dev_switches_parse_and_reject_unknown_keys():
    k = _dev._parse("row_sweep=1, encoder_stream=0,unet_wgrad_stream=off,lib_dir=/tmp/x")
    assert k.row_sweep is True and k.encoder_stream is False and k.unet_wgrad_stream is False and k.lib_dir == "/tmp/x"
    d = _dev._parse("")
    assert d.one_pass_backward is True and d.encoder_stream is True and d.native_unet is True and d.wgrad_stream is True
    with pytest.raises(ValueError):
        _dev._parse("no_such_switch=1")


def test_close_is_idempotent_and_leaves_the_collector_thawed():
    from gaussianavatar_amd.avatar_model import AvatarModel, default_params
    m = AvatarModel(*default_params(batch_size=1, num_points=500, image_width=32, image_height=32, num_frames=2),
                    train=True, device="cpu")
    m.training_setup()
    m.close()
    m.close()
    assert gc.get_freeze_count() == 0


def test_training_loader_budget_defaults_and_override():
    from gaussianavatar_amd.avatar_model import _DeviceLoader
    assert _DeviceLoader([], "cpu").budget == 64 << 30                      # host memory: the old fixed figure
    assert _DeviceLoader([], "cpu", budget_gb=0.5).budget == 1 << 29
    assert _DeviceLoader([], torch.device("cpu"), budget_gb=0).budget == 0   # never resident: keeps streaming
