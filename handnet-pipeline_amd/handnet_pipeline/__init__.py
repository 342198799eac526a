"""Drop-in for the reference package `handnet_pipeline`."""
from .handnet_pipeline import HandNet, HandNetPipeline  # noqa: F401
