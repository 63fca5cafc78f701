"""/root/reference/model/modules.py — the modules on the hot path: GeomConvLayers (:114-137),
UnetNoCond5DS (:185-232), ShapeDecoder (:508-582), uv_to_grid (:745-754)."""
from gaussianavatar_amd.network import GeomConvLayers, ShapeDecoder, UnetNoCond5DS, uv_to_grid  # noqa: F401
