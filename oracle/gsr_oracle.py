"""ctypes front-end of the CPU rasterizer ORACLE (oracle/gsr_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package (gaussianavatar_amd/) never imports it.

PARITY UNPINNED (see the header of gsr_oracle.c): restates SURVEY.md Appendix A for the
un-vendored `diff_gaussian_rasterization` dependency used at
/root/reference/gaussian_renderer/__init__.py:6,21-48.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (a few hundred ms)."""
    outs = [os.path.join(_BUILD, f"libgsr_oracle_{p}.so") for p in ("f32", "f64", "f32_fma")]
    src = os.path.join(_HERE, "gsr_oracle.c")
    fresh = all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs)
    if fresh and not force:
        return
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))



def _cam_type(real):
    class Cam(ctypes.Structure):
        _fields_ = [
            ("W", ctypes.c_int),
            ("H", ctypes.c_int),
            ("tanfovx", real),
            ("tanfovy", real),
            ("scale_modifier", real),
            ("bg", ctypes.c_void_p),
            ("view", ctypes.c_void_p),
            ("proj", ctypes.c_void_p),
        ]

    return Cam


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class RasterOracle:
    """CPU restatement of the rasterizer forward/backward. `f64=True` selects the double
    build (gradient reference); the default float build fixes the integer outputs."""

    def __init__(self, f64: bool = False, fma: bool = False):
        """fma=True (float32 only): the build with FMA contraction allowed (-ffp-contract=fast -mfma), for the
        contraction-sensitivity measurement; every parity test uses the contraction-off builds."""
        build()
        assert not (f64 and fma)
        name = "libgsr_oracle_f64.so" if f64 else ("libgsr_oracle_f32_fma.so" if fma else "libgsr_oracle_f32.so")
        self.lib = ctypes.CDLL(os.path.join(_BUILD, name))
        self.dtype = np.float64 if f64 else np.float32
        self.creal = ctypes.c_double if f64 else ctypes.c_float
        self.Cam = _cam_type(self.creal)
        assert self.lib.gsro_real_bytes() == np.dtype(self.dtype).itemsize
        self.lib.gsro_bin.restype = ctypes.c_int64

    def _arr(self, a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a, dtype=self.dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def sh_colors(self, means3D, shs, sh_degree, campos):
        """View-dependent colours from SH coefficients shs [P,M,3] -> (colors [P,3], clamped [P,3])."""
        P = int(np.asarray(means3D).shape[0])
        shs = self._arr(shs)
        M = int(shs.shape[1])
        assert shs.shape == (P, M, 3) and M >= (int(sh_degree) + 1) ** 2
        means3D = self._arr(means3D, (P, 3))
        campos = self._arr(campos, (3,))
        colors = np.zeros((P, 3), self.dtype)
        clamped = np.zeros((P, 3), np.uint8)
        self.lib.gsro_sh_forward(P, M, int(sh_degree), _p(means3D), _p(campos), _p(shs), _p(colors),
                                 _p(clamped))
        return colors, clamped

    def sh_backward(self, means3D, shs, sh_degree, campos, clamped, dL_dcolors):
        """-> (dL_dsh [P,M,3], view-direction term of dL_dmeans3D [P,3])."""
        P = int(np.asarray(means3D).shape[0])
        shs = self._arr(shs)
        M = int(shs.shape[1])
        means3D = self._arr(means3D, (P, 3))
        campos = self._arr(campos, (3,))
        clamped = np.ascontiguousarray(clamped, dtype=np.uint8)
        g = self._arr(dL_dcolors, (P, 3))
        dsh = np.zeros((P, M, 3), self.dtype)
        dmean = np.zeros((P, 3), self.dtype)
        self.lib.gsro_sh_backward(P, M, int(sh_degree), _p(means3D), _p(campos), _p(shs), _p(clamped),
                                  _p(g), _p(dsh), _p(dmean))
        return dsh, dmean

    def forward(self, means3D, colors, opacities, scales=None, rotations=None, cov3D_precomp=None,
                *, viewmatrix, projmatrix, bg, W, H, tanfovx, tanfovy, scale_modifier=1.0,
                shs=None, sh_degree=0, campos=None):
        P = int(np.asarray(means3D).shape[0])
        means3D = self._arr(means3D, (P, 3))
        sh_state = None
        if shs is not None:
            assert colors is None, "exactly one of colors / shs"
            shs = self._arr(shs)
            campos = self._arr(campos, (3,))
            colors, clamped = self.sh_colors(means3D, shs, sh_degree, campos)
            sh_state = dict(shs=shs, sh_degree=int(sh_degree), campos=campos, clamped=clamped)
        colors = self._arr(colors, (P, 3))
        opacities = self._arr(opacities, (P,))
        scales = self._arr(scales, (P, 3)) if scales is not None else None
        rotations = self._arr(rotations, (P, 4)) if rotations is not None else None
        cov3D_precomp = self._arr(cov3D_precomp, (P, 6)) if cov3D_precomp is not None else None
        assert (scales is None) == (rotations is None)
        assert (scales is None) != (cov3D_precomp is None)
        view = self._arr(viewmatrix, (16,))
        proj = self._arr(projmatrix, (16,))
        bg = self._arr(bg, (3,))
        cam = self.Cam(int(W), int(H), float(tanfovx), float(tanfovy), float(scale_modifier),
                       _p(bg), _p(view), _p(proj))
        st = dict(P=P, W=int(W), H=int(H), cam=cam, _keep=(bg, view, proj), bg=bg,
                  means3D=means3D, colors=colors, opacities=opacities, scales=scales,
                  rotations=rotations, cov3D_precomp=cov3D_precomp, sh=sh_state)
        f = self.dtype
        st["depth"] = np.zeros(P, f)
        st["xy"] = np.zeros((P, 2), f)
        st["conic_opacity"] = np.zeros((P, 4), f)
        st["cov3d"] = np.zeros((P, 6), f)
        st["radii"] = np.zeros(P, np.int32)
        st["rect"] = np.zeros((P, 4), np.int32)
        st["tiles_touched"] = np.zeros(P, np.uint32)
        self.lib.gsro_preprocess(P, _p(means3D), _p(scales), _p(rotations), _p(cov3D_precomp),
                                 _p(opacities), ctypes.byref(cam), _p(st["depth"]), _p(st["xy"]),
                                 _p(st["conic_opacity"]), _p(st["cov3d"]), _p(st["radii"]),
                                 _p(st["rect"]), _p(st["tiles_touched"]))
        gx, gy = (W + 15) // 16, (H + 15) // 16
        T = gx * gy
        D = int(st["tiles_touched"].astype(np.int64).sum())
        st["ranges"] = np.zeros((T, 2), np.uint32)
        st["point_list"] = np.zeros(max(D, 1), np.uint32)
        got = self.lib.gsro_bin(P, _p(st["rect"]), _p(st["tiles_touched"]), _p(st["depth"]),
                                int(W), int(H), _p(st["ranges"]), _p(st["point_list"]),
                                ctypes.c_int64(max(D, 1)))
        assert got == D, (got, D)
        st["D"] = D
        st["point_list"] = st["point_list"][:D] if D else st["point_list"][:0]
        st["color"] = np.zeros((3, H, W), f)
        st["final_T"] = np.zeros(H * W, f)
        st["n_contrib"] = np.zeros(H * W, np.uint32)
        pl = st["point_list"] if D else np.zeros(1, np.uint32)
        self.lib.gsro_render(int(W), int(H), _p(st["ranges"]), _p(pl), _p(st["xy"]),
                             _p(st["conic_opacity"]), _p(colors), _p(bg), _p(st["color"]),
                             _p(st["final_T"]), _p(st["n_contrib"]))
        return st

    def backward(self, st, dL_dout):
        P, W, H = st["P"], st["W"], st["H"]
        f = self.dtype
        g = self._arr(dL_dout, (3, H, W))
        out = dict(dmean2D=np.zeros((P, 2), f), dconic=np.zeros((P, 3), f),
                   dopacity=np.zeros(P, f), dcolors=np.zeros((P, 3), f))
        pl = st["point_list"] if st["D"] else np.zeros(1, np.uint32)
        self.lib.gsro_render_backward(P, W, H, _p(st["ranges"]), _p(pl), _p(st["xy"]),
                                      _p(st["conic_opacity"]), _p(st["colors"]), _p(st["bg"]),
                                      _p(st["final_T"]), _p(st["n_contrib"]), _p(g),
                                      _p(out["dmean2D"]), _p(out["dconic"]), _p(out["dopacity"]),
                                      _p(out["dcolors"]))
        out["dmeans3D"] = np.zeros((P, 3), f)
        out["dcov3D"] = np.zeros((P, 6), f)
        out["dscales"] = np.zeros((P, 3), f)
        out["drots"] = np.zeros((P, 4), f)
        self.lib.gsro_preprocess_backward(P, _p(st["means3D"]), _p(st["scales"]),
                                          _p(st["rotations"]), _p(st["cov3d"]), _p(st["radii"]),
                                          ctypes.byref(st["cam"]), _p(out["dmean2D"]),
                                          _p(out["dconic"]), _p(out["dmeans3D"]),
                                          _p(out["dcov3D"]), _p(out["dscales"]), _p(out["drots"]))
        if st.get("sh") is not None:
            sh = st["sh"]
            M = int(sh["shs"].shape[1])
            out["dsh"] = np.zeros((P, M, 3), f)
            self.lib.gsro_sh_backward(P, M, sh["sh_degree"], _p(st["means3D"]), _p(sh["campos"]),
                                      _p(sh["shs"]), _p(sh["clamped"]), _p(out["dcolors"]),
                                      _p(out["dsh"]), _p(out["dmeans3D"]))
        # API shape of the means2D gradient: [P,3] with a zero z column
        out["dmeans2D"] = np.concatenate([out["dmean2D"], np.zeros((P, 1), f)], axis=1)
        return out

    def tile_lists(self, st):
        """Per-tile depth-sorted Gaussian index lists as a python list of arrays."""
        return [st["point_list"][s:e] for s, e in st["ranges"]]
