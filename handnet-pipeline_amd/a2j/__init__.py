"""Drop-in for the reference package `a2j` (only the inference entry point is provided)."""
from .a2j import A2JModel, convert_joints  # noqa: F401
