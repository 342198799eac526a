"""Reference-layout parameter containers.

The drop-in classes (a2j.a2j.A2JModel, fcos_utils.fcos.FCOS, handnet_pipeline.HandNet)
must accept the reference's checkpoints through the ordinary nn.Module protocol
(`load_state_dict(ckpt["model"], strict=False)`, `.cuda()`, `.eval()`), so they own an
nn.Module tree whose parameter / buffer names and shapes equal the reference's
(SURVEY A.6).  The tree holds STATE ONLY -- there is no torch forward; compute happens in
the HIP engines built from that state.
"""
from __future__ import annotations

import torch
from torch import nn

_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked", "all_anchors", "thres")


class StateTree(nn.Module):
    """A bare namespace node; children/parameters are attached by build_state_tree."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("StateTree holds reference-layout weights only; run the HIP engine instead")


def build_state_tree(state_dict) -> StateTree:
    """Create a module tree whose state_dict() has exactly the keys/shapes of `state_dict`."""
    root = StateTree()
    for key, value in state_dict.items():
        parts = key.split(".")
        node = root
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, StateTree())
            node = getattr(node, p)
        leaf = parts[-1]
        t = value.detach().clone()
        if leaf in _BUFFER_SUFFIXES or not t.is_floating_point():
            node.register_buffer(leaf, t)
        else:
            node.register_parameter(leaf, nn.Parameter(t, requires_grad=False))
    return root


class EngineOwner(nn.Module):
    """nn.Module that lazily (re)builds a HIP engine from its current state."""

    def __init__(self):
        super().__init__()
        self._engine = None

    def _invalidate(self):
        self._engine = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate_all()
        return out

    def _invalidate_all(self):
        for m in self.modules():
            if isinstance(m, EngineOwner):
                m._invalidate()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate_all()
        return out

    def _device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        for b in self.buffers():
            return b.device
        return torch.device("cpu")

    def _require_gpu(self) -> torch.device:
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError(
                f"{type(self).__name__} runs on hand-written HIP kernels only: move it to the GPU with .cuda() "
                "(there is no CPU fallback in handnet-pipeline_amd)")
        return dev
