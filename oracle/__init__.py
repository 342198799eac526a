"""CPU oracle for the FCOS -> crop -> A2J hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain PyTorch-CPU fp32 / numpy / C, the algorithm of the
reference (IRVLUTD/handnet-pipeline) for the path named in BASELINE.json.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it -- as the
checker, never as the thing measured or shipped.  The product (handnet-pipeline_amd/)
never imports it and has no CPU fallback.

Pinning status
  * A2J (a2j_ref.py): pinned -- tests/golden/a2j_*.npz were produced by importing the
    reference's own a2j/a2j.py, a2j/resnet.py, a2j/anchor.py in the build container
    (tests/golden/make_golden.py) and the restatement reproduces them.
  * HandNet glue (handnet_ref.py) and the in-repo FCOS pieces (heads, anchors, box
    decode, score/threshold, dict assembly: fcos_ref.py): pinned the same way
    (tests/golden/fcos_*.npz, handnet_*.npz).
  * torchvision 0.11.3 pieces that the reference only CALLS (resnet34 body, FrozenBN,
    FPN, GeneralizedRCNNTransform, batched_nms/nms): torchvision is absent from
    /root/reference and from this image, the reference holds no tests or fixtures for
    them => PARITY UNPINNED for those functions; they are restated from the published
    torchvision-0.11.3 algorithm and anchored on the reference call sites
    (fcos_utils/fcos.py:476,505,635,709,737).  Partial pin: stem + layer1-3 of the trunk reproduce the
    reference's in-tree a2j/resnet.py ResNet(BasicBlock) (tests/golden/resnet34_intree.npz).
    Which upstream file each restated function follows (torchvision tag v0.11.3, pinned by scripts/init_env.sh:25):
      fcos_ref.nms / oracle/nms_ref.c ......... torchvision/csrc/ops/cpu/nms_kernel.cpp (nms_kernel_impl: areas as a tensor op,
                                                descending sort, greedy loop, `ovr > iou_threshold` with a double threshold)
      fcos_ref.batched_nms .................... torchvision/ops/boxes.py (batched_nms -> _batched_nms_vanilla when
                                                boxes.numel() > 4000, else _batched_nms_coordinate_trick)
      fcos_ref.transform / resized_size ....... torchvision/models/detection/transform.py (GeneralizedRCNNTransform.normalize,
                                                _resize_image_and_masks, batch_images with size_divisible = 32)
      fcos_ref.fpn ............................ torchvision/ops/feature_pyramid_network.py (FeaturePyramidNetwork.forward)
      fcos_ref.body / _basic_block / _frozen_bn  torchvision/models/resnet.py (BasicBlock, ResNet._forward_impl),
                                                torchvision/ops/misc.py (FrozenBatchNorm2d)
    Anchors that narrow the unpinned surface without torchvision (round 3): greedy NMS is pinned on its DEFINITION
    (tests/nms_property.py: on tie-free boxes the kept set is unique; checked in fp64 for the C restatement and for
    the HIP kernel independently of each other), the FPN top-down wiring on hand-written fp64 loops
    (tests/test_oracle_cpu.py), FrozenBN and the whole trunk on HF transformers' independent copies.  What remains
    unpinned is the operation ORDER inside the fp32 IoU expression and torchvision's choice of interpolation flags.
  * Pose2Mesh lifter (pose2mesh_ref.py): arithmetic pinned (tests/golden/pose2mesh_forward.npz, reference
    modules imported) on a synthetic mesh hierarchy -- the MANO files the real graphs derive from are absent.
"""
