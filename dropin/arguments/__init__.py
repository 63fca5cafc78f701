"""Command-line parameter groups with the flags and defaults of /root/reference/arguments/__init__.py:55-165,
so `train.py` / `eval.py` / `render_novel_pose.py` parse the same command lines (`-s`, `-m`, `-w`,
`--train_stage`, ...). No pytorch3d: the canonical-pose constants are written out directly.

A group is declared as a plain table of (name, default); a leading underscore in the reference's
attribute names (`_source_path`) means "also register the one-letter flag" — kept here as the
`short` set. `sentinel=True` registers every default as None so that `get_combined_args` can tell
"given on the command line" from "take it from the training run's cfg_args file".
"""
from __future__ import annotations

import math
import os
import sys
from argparse import ArgumentParser, Namespace

import torch


class GroupParams:
    pass


class ParamGroup:
    def __init__(self, parser: ArgumentParser, name: str, fill_none: bool = False):
        group = parser.add_argument_group(name)
        for attr, default in list(vars(self).items()):
            short = attr.startswith("_")
            key = attr[1:] if short else attr
            flags = ["--" + key] + (["-" + key[0]] if short else [])
            kind = type(default)
            value = None if fill_none else default
            if kind is bool:
                group.add_argument(*flags, default=value, action="store_true")
            else:
                group.add_argument(*flags, default=value, type=kind)

    def extract(self, args):
        out = GroupParams()
        mine = vars(self)
        for key, value in vars(args).items():
            if key in mine or ("_" + key) in mine:
                setattr(out, key, value)
        return out


# canonical pose: legs spread by 30 degrees about z (arguments/__init__.py:44-53)
leg_angle = 30
smplx_cpose_param = torch.zeros(1, 165)
smplx_cpose_param[:, 5] = leg_angle / 180 * math.pi
smplx_cpose_param[:, 8] = -leg_angle / 180 * math.pi
smpl_cpose_param = torch.zeros(1, 72)
smpl_cpose_param[:, 5] = leg_angle / 180 * math.pi
smpl_cpose_param[:, 8] = -leg_angle / 180 * math.pi


class ModelParams(ParamGroup):
    def __init__(self, parser, sentinel=False):
        cwd = os.getcwd()
        self._source_path = ""
        self._model_path = ""
        self.project_path = cwd
        self.smpl_model_path = cwd + "/assets/smpl_files/smpl"
        self.smplx_model_path = cwd + "/assets/smpl_files/smplx"
        self.test_folder = cwd + "/assets/test_pose"
        self.stage1_out_path = ""
        self.save_epoch = 30
        self.train_stage = 1
        self.dataset_type = "peeplesnapshot"
        self.smpl_gender = "neutral"
        self.smpl_type = "smpl"
        self.no_mask = 0
        self.fixed_inp = 0
        self.train_mode = 0
        self.cam_static = 1
        self._white_background = True
        self.bullet_pose_list = [112, 217, 755]
        self.batch_size = 2
        self.query_posmap_size = 512
        self.inp_posmap_size = 128
        super().__init__(parser, "Loading Parameters", sentinel)

    def extract(self, args):
        g = super().extract(args)
        g.source_path = os.path.abspath(g.source_path)
        return g


class NetworkParams(ParamGroup):
    def __init__(self, parser):
        self.c_pose = 64
        self.c_geom = 64
        self.hsize = 128
        self.nf = 32
        self.up_mode = "upconv"
        self.use_dropout = 0
        self.pos_encoding = 0
        self.num_emb_freqs = 6
        self.posemb_incl_input = 0
        self.geom_layer_type = "conv"
        self.gaussian_kernel_size = 5
        super().__init__(parser, "Network Parameters")


class OptimizationParams(ParamGroup):
    def __init__(self, parser):
        self.epochs = 200
        self.position_lr_init = 0.00016
        self.position_lr_final = 0.0000016
        self.position_lr_delay_mult = 0.01
        self.position_lr_max_steps = self.epochs
        self.feature_lr = 0.0025
        self.opacity_lr = 0.05
        self.scaling_lr = 0.005
        self.rotation_lr = 0.001
        self.percent_dense = 0.01
        self.lambda_dssim = 0.2
        self.lambda_scale = 3e-2
        self.lambda_lpips = 0.2
        self.lambda_aiap = 0.1
        self.lambda_color = 3e-2
        self.lambda_pose = 10
        self.lambda_rgl = 1e1
        self.log_iter = 2000
        self.lpips_start_iter = 30
        self.pose_op_start_iter = 1800
        self.lr_net = 3e-3
        self.lr_geomfeat = 5e-4
        # NB (reference quirk, kept): the milestones are derived from the DEFAULT epoch count, and
        # argparse applies `type=list` to a command-line value for this flag
        self.sched_milestones = [int(self.epochs / 3), int(self.epochs * 2 / 3)]
        super().__init__(parser, "Optimization Parameters")


def get_combined_args(parser: ArgumentParser):
    """Command line over the `cfg_args` file the training run left in its model_path
    (arguments/__init__.py:145-165)."""
    args_cmdline = parser.parse_args(sys.argv[1:])
    cfg_text = "Namespace()"
    try:
        cfg_path = os.path.join(args_cmdline.model_path, "cfg_args")
        print("Looking for config file in", cfg_path)
        with open(cfg_path) as f:
            print("Config file found: {}".format(cfg_path))
            cfg_text = f.read()
    except TypeError:
        print("Config file not found at")
    merged = vars(eval(cfg_text, {"Namespace": Namespace})).copy()
    for k, v in vars(args_cmdline).items():
        if v is not None:
            merged[k] = v
    return Namespace(**merged)
