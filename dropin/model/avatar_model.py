"""/root/reference/model/avatar_model.py -> gaussianavatar_amd.avatar_model (same class, same methods)."""
from gaussianavatar_amd.avatar_model import AvatarModel  # noqa: F401
