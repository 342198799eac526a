"""Drop-in for the reference package `a2j` (only the inference entry points are provided)."""
from .a2j import A2JModel, A2JModelLightning, convert_joints  # noqa: F401
