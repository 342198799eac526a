"""Oracle for the Pose2Mesh lifter: CPU restatement of
pose2mesh/lib/models/pose2mesh_net.py:9-24 (FlatPose2Mesh), posenet.py:12-88 (LinearModel, eval),
meshnet.py:79-117 (Pose2Mesh.forward, 'mano' configuration) and backbones/cheby_graph_conv.py:5-42.
Test infrastructure only.  Pinned by tests/golden/pose2mesh_forward.npz (reference modules imported in the
build container on a synthetic mesh hierarchy; see tests/golden/make_golden_p2m.py).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

CL_F = [(5, 32, 64, 64), (64, 128, 256), (256, 256, 256), (256, 256, 256), (256, 256, 256), (256, 128, 128),
        (128, 64, 3)]   # meshnet.py:22-27 ('mano')
CL_K = 3               # meshnet.py:21
BN_EPS = 1e-5


def load_graphs(npz):
    """Laplacians of the fixture (finest first, joint graph last) as scipy CSR matrices."""
    out = []
    for i in range(int(npz["num_levels"])):
        shape = tuple(int(v) for v in npz[f"L{i}_shape"])
        out.append(sp.csr_matrix((npz[f"L{i}_data"], npz[f"L{i}_indices"], npz[f"L{i}_indptr"]), shape=shape))
    return out


def _bn_eval(x, sd, name):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], False, 0.0, BN_EPS)


def posenet(x, sd, p="pose_lifter."):
    """LinearModel.forward, eval (posenet.py:78-88; Linear.forward :27-41; dropout is the identity)."""
    y = F.linear(x, sd[p + "w1.weight"], sd[p + "w1.bias"])
    for st in range(2):
        q = f"{p}linear_stages.{st}."
        z = F.relu(_bn_eval(y, sd, q + "batch_norm1"))
        z = F.linear(z, sd[q + "w1.weight"], sd[q + "w1.bias"])
        z = F.relu(_bn_eval(z, sd, q + "batch_norm2"))
        z = F.linear(z, sd[q + "w2.weight"], sd[q + "w2.bias"])
        y = y + z
    return F.linear(y, sd[p + "w2.weight"], sd[p + "w2.bias"])


def _to_torch_sparse(L):
    c = L.tocoo()
    idx = torch.from_numpy(np.vstack((c.row, c.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(c.data.astype(np.float32)), c.shape).coalesce()


def graph_conv_cheby(x, weight, bias, bn, L, K):
    """cheby_graph_conv.py:5-42.  x [B,V,Fin]; features are ordered (fin, k) in the Linear's input."""
    B, V, Fin = x.shape
    x0 = x.permute(1, 2, 0).contiguous().view(V, Fin * B)
    xs = [x0]
    if K > 1:
        x1 = torch.sparse.mm(L, x0)
        xs.append(x1)
    for _ in range(2, K):
        x2 = 2 * torch.sparse.mm(L, x1) - x0
        xs.append(x2)
        x0, x1 = x1, x2
    t = torch.stack(xs, 0).view(K, V, Fin, B).permute(3, 1, 2, 0).contiguous().view(B * V, Fin * K)
    t = F.linear(t, weight, bias)
    if bn is not None:
        t = F.batch_norm(t, bn[0], bn[1], bn[2], bn[3], False, 0.0, BN_EPS)
    return t.view(B, V, -1)


def meshnet(x, sd, graphs, p="pose2mesh."):
    """Pose2Mesh.forward (meshnet.py:79-117).  graphs: full hierarchy; level [-2] is dropped (meshnet.py:37)."""
    Ls = [_to_torch_sparse(g) for g in graphs]
    del Ls[-2]
    B = x.shape[0]
    x = x.view(-1, Ls[-1].shape[0], CL_F[0][0])
    cl_i = 0
    nblk = len(CL_F)
    for i in range(nblk):
        input_x = x
        for li in range(len(CL_F[i]) - 1):
            ldx = -(i + 1) + (1 if i == nblk - 1 else 0)
            last = i == nblk - 1 and li == len(CL_F[i]) - 2
            bn = None if last else tuple(sd[f"{p}bn.{cl_i}.{k}"] for k in ("running_mean", "running_var", "weight", "bias"))
            x = graph_conv_cheby(x, sd[f"{p}cl.{cl_i}.weight"], sd[f"{p}cl.{cl_i}.bias"], bn, Ls[ldx], CL_K)
            if not last:
                x = F.relu(x)
            cl_i += 1
        if i == 0:      # joints -> coarsest mesh level through one dense layer (meshnet.py:101-103)
            x = F.linear(x.reshape(B, -1), sd[p + "fc.weight"], sd[p + "fc.bias"]).view(B, Ls[-2].shape[0], CL_F[1][0])
        elif i < nblk - 2:
            input_x = F.interpolate(input_x, size=x.shape[2], mode="linear")   # along the FEATURE axis (meshnet.py:106)
            x = input_x + x
            x = x.permute(0, 2, 1).contiguous()
            x = F.interpolate(x, scale_factor=2)                                # nn.Upsample(scale_factor=2), nearest
            x = x.permute(0, 2, 1).contiguous()
        elif i == nblk - 2:
            input_x = F.interpolate(input_x, size=x.shape[2], mode="linear")
            x = input_x + x
    return x


def pose2mesh_forward(pose2d, sd, graphs):
    """FlatPose2Mesh.forward (pose2mesh_net.py:17-24): pose2d [B,J,2] -> (cam_mesh [B,V,3], pose3d [B,J,3])."""
    with torch.no_grad():
        B, J = pose2d.shape[:2]
        pose3d = posenet(pose2d.reshape(B, -1), sd).reshape(-1, J, 3)
        comb = torch.cat((pose2d, pose3d / 1000), dim=2)
        return meshnet(comb, sd, graphs), pose3d


# ---------------------------------------------------------------------------------------------------------------------------
# The live caller's glue between the pose network and the lifter (ros_demo.py:148-157), restated function by function.
# cv2.getAffineTransform (absent from this image) is the exact solution of the three-point correspondence: numpy solves it.
# ---------------------------------------------------------------------------------------------------------------------------
INPUT_SHAPE = (384, 288)    # cfg.MODEL.input_shape, pose2mesh/lib/core/config.py:52


def get_bbox(joint_img):
    """pose2mesh/lib/coord_utils.py:21-40"""
    import numpy as np
    x_img, y_img = joint_img[:, 0], joint_img[:, 1]
    xmin, ymin, xmax, ymax = min(x_img), min(y_img), max(x_img), max(y_img)
    x_center = (xmin + xmax) / 2.
    width = xmax - xmin
    xmin, xmax = x_center - 0.5 * width, x_center + 0.5 * width
    y_center = (ymin + ymax) / 2.
    height = ymax - ymin
    ymin, ymax = y_center - 0.5 * height, y_center + 0.5 * height
    return np.array([xmin, ymin, xmax - xmin, ymax - ymin]).astype(np.float32)


def process_bbox(bbox, aspect_ratio=None, scale=1.0):
    """pose2mesh/lib/coord_utils.py:42-66"""
    import numpy as np
    x, y, w, h = bbox
    x1, y1, x2, y2 = x, y, x + (w - 1), y + (h - 1)
    if w * h > 0 and x2 >= x1 and y2 >= y1:
        bbox = np.array([x1, y1, x2 - x1, y2 - y1])
    else:
        return None
    w, h = bbox[2], bbox[3]
    c_x, c_y = bbox[0] + w / 2., bbox[1] + h / 2.
    if aspect_ratio is None:
        aspect_ratio = INPUT_SHAPE[1] / INPUT_SHAPE[0]
    if w > aspect_ratio * h:
        h = w / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    bbox[2], bbox[3] = w * scale, h * scale
    bbox[0], bbox[1] = c_x - bbox[2] / 2., c_y - bbox[3] / 2.
    return bbox


def _affine_transform_rot0(center, scale, output_size):
    """pose2mesh/lib/aug_utils.py:140-173 with rot = 0, shift = 0, inv = 0"""
    import numpy as np
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    src_dir = [0.0, src_w * -0.5]                       # get_dir at rot_rad = 0 (:188-195)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = center + np.array(src_dir, dtype=np.float32)
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    third = lambda a, b: b + np.array([-(a - b)[1], (a - b)[0]], dtype=np.float32)     # get_3rd_point (:182-184)
    src[2, :] = third(src[0, :], src[1, :])
    dst[2, :] = third(dst[0, :], dst[1, :])
    a = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)             # [x y 1] T^t = [x' y']
    return np.linalg.solve(a, dst.astype(np.float64)).T                                # 2 x 3, as cv2.getAffineTransform returns


def lifter_input(joint_input):
    """ros_demo.py:148-157 (predict_mesh up to the model call): image-(u,v) joints [J,2] -> the standardised [J,2] the lifter
    is fed, or None when the caller would skip the frame (process_bbox rejects the box)."""
    import numpy as np
    joint_input = np.asarray(joint_input, dtype=np.float32)
    bbox = get_bbox(joint_input)
    bbox2 = process_bbox(bbox.copy())
    if bbox2 is None:
        return None
    # j2d_processing(joint_input.copy(), (input_shape[1], input_shape[0]), bbox2, 0, 0, None)  (aug_utils.py:51-64)
    center = np.zeros((2,), dtype=np.float32)                                          # get_center_scale (coord_utils.py:7-18)
    center[0], center[1] = bbox2[0] + bbox2[2] * 0.5, bbox2[1] + bbox2[3] * 0.5
    scale = np.array([bbox2[2] * 1.0, bbox2[3] * 1.0], dtype=np.float32)
    trans = _affine_transform_rot0(center, scale, (INPUT_SHAPE[1], INPUT_SHAPE[0]))
    kp = joint_input.copy()
    for i in range(kp.shape[0]):
        kp[i, :2] = np.dot(trans, np.array([kp[i, 0], kp[i, 1], 1.]).T)[:2]            # affine_transform (:176-179)
    kp = kp.astype("float32")
    joint_img = kp[:, :2]
    joint_img /= np.array([[INPUT_SHAPE[1], INPUT_SHAPE[0]]])
    mean, std = np.mean(joint_img, axis=0), np.std(joint_img, axis=0)
    return ((joint_img.copy() - mean) / std).astype(np.float32)
