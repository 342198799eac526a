"""Drop-in for the reference's `models` package of pose2mesh/lib (only the lifter used by ros_demo.py:142)."""
from . import pose2mesh_net  # noqa: F401
