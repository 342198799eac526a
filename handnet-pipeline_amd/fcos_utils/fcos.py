"""Drop-in `fcos_utils.fcos.FCOS` running on MI355X HIP kernels.

Same constructor arguments, state_dict layout and eval-mode call contract as the
reference class (fcos_utils/fcos.py:398-767): `model(images: list of [3,H,W] in 0..1)`
returns one dict per image with `boxes [k,4]` (original-image pixels, unclipped),
`scores [k]`, `labels [k] int64`, `sides [k] int64`, `feature_idx [k] float32`, sorted by
descending score; with `ext=True` (the class default, used by trainval_net_fcos.py --test-only) the dicts hold
`dxdymags [k,3]`, `contacts [k] int64` and `sides` instead of `feature_idx` (fcos.py:637-647).  As in the reference, `score_thresh / nms_thresh / topk_candidates /
detections_per_img` are accepted and IGNORED: post-processing hard-codes score > 0.7 and
NMS IoU 0.3 (fcos.py:600,635).  Training (targets / losses) is out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from hn_amd import synth
from hn_amd.fcos_engine import FCOSEngine
from hn_amd.state import EngineOwner, build_state_tree


class FCOS(EngineOwner):
    def __init__(self, num_classes: int, ext: bool = True, min_size: int = 800, max_size: int = 1333,
                 image_mean: Optional[List[float]] = None, image_std: Optional[List[float]] = None,
                 anchor_generator=None, head=None, center_sampling_radius: float = 1.5,
                 score_thresh: float = 0.2, nms_thresh: float = 0.6, detections_per_img: int = 100,
                 topk_candidates: int = 1000):
        super().__init__()
        if anchor_generator is not None or head is not None:
            # (the reference accepts torch modules here, fcos.py:483-494; this class runs fixed HIP layer graphs --
            # listed under "deviations" in INTEGRATION.md)
            raise NotImplementedError("custom anchor_generator / head modules are not supported")
        # forwarded to the transform like the reference does (fcos.py:501-505)
        self.image_mean = [0.485, 0.456, 0.406] if image_mean is None else [float(v) for v in image_mean]
        self.image_std = [0.229, 0.224, 0.225] if image_std is None else [float(v) for v in image_std]
        self.ext = ext
        self.num_classes = num_classes
        self.min_size, self.max_size = min_size, max_size
        # stored-and-ignored, exactly like the reference (fcos.py:507-511)
        self.center_sampling_radius = center_sampling_radius
        self.score_thresh, self.nms_thresh = score_thresh, nms_thresh
        self.detections_per_img, self.topk_candidates = detections_per_img, topk_candidates
        tree = build_state_tree(synth.make_fcos_state_dict(seed=0, num_classes=num_classes, ext=ext))
        for name, child in tree.named_children():
            self.add_module(name, child)

    def engine(self) -> FCOSEngine:
        dev = self._require_gpu()
        if self._engine is None:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._engine = FCOSEngine(sd, self.num_classes, device=dev, min_size=self.min_size,
                                      max_size=self.max_size, ext=self.ext, image_mean=self.image_mean,
                                      image_std=self.image_std)
        return self._engine

    def forward(self, images: List[torch.Tensor], targets=None) -> List[Dict[str, torch.Tensor]]:
        if self.training or targets is not None:
            raise NotImplementedError("training (fcos.py:543-570 losses) is outside the inference hot path")
        if len({tuple(i.shape) for i in images}) != 1:
            # torchvision batch_images (fcos.py:702-709): every image is resized on its own, padded to the common
            # canvas, and its boxes are scaled back by its own ratios
            batch = [i.float() for i in images]
        else:
            batch = torch.stack([i.float() for i in images])
        if self.ext:
            det, _, contacts, dxdymags = self.engine().detect_ext(batch)
        else:
            det, _ = self.engine().detect(batch)
        counts = det.count.cpu().tolist()  # the list-of-dicts contract needs lengths on the host
        out = []
        for i, k in enumerate(counts):
            if self.ext:  # fcos.py:637-647
                out.append({
                    "boxes": det.boxes[i, :k].clone(),
                    "scores": det.scores[i, :k].clone(),
                    "labels": det.labels[i, :k].to(torch.int64),
                    "dxdymags": dxdymags[i, :k].clone(),
                    "contacts": contacts[i, :k].to(torch.int64),
                    "sides": det.sides[i, :k].to(torch.int64),
                })
                continue
            out.append({
                "boxes": det.boxes[i, :k].clone(),
                "scores": det.scores[i, :k].clone(),
                "labels": det.labels[i, :k].to(torch.int64),
                "sides": det.sides[i, :k].to(torch.int64),
                "feature_idx": det.level[i, :k].to(torch.float32),
            })
        return out
