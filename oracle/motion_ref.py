"""CPU restatement of the reference's post-processing of generated motions (TEST INFRASTRUCTURE ONLY).

Follows, for the HumanML3D 263-d representation with 22 joints:
  * de-normalisation `motion * std + mean`                                         tools/visualization.py:89
  * recover_root_rot_pos                                                           utils/motion_process.py:362-382
  * recover_from_ric                                                               utils/motion_process.py:403-416
  * qinv / qrot                                                                    utils/quaternion.py:16-20,54-73
  * motion_temporal_filter = scipy gaussian_filter(sigma, mode="nearest") per joint coordinate  utils/utils.py:125-130
Pinned by tests/golden/motion_post.npz (outputs of the reference functions, oracle/make_golden.py::case_motion_post).
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def qinv(q):
    mask = torch.ones_like(q)
    mask[..., 1:] = -mask[..., 1:]
    return q * mask


def qrot(q, v):
    shape = list(v.shape)
    q = q.contiguous().view(-1, 4)
    v = v.contiguous().view(-1, 3)
    qvec = q[:, 1:]
    uv = torch.cross(qvec, v, dim=1)
    uuv = torch.cross(qvec, uv, dim=1)
    return (v + 2 * (q[:, :1] * uv + uuv)).view(shape)


def recover_root_rot_pos(data):
    rot_vel = data[..., 0]
    ang = torch.zeros_like(rot_vel)
    ang[..., 1:] = rot_vel[..., :-1]
    ang = torch.cumsum(ang, dim=-1)                       # Y-axis rotation from its velocity (:364-367)
    quat = torch.zeros(data.shape[:-1] + (4,))
    quat[..., 0] = torch.cos(ang)
    quat[..., 2] = torch.sin(ang)
    pos = torch.zeros(data.shape[:-1] + (3,))
    pos[..., 1:, [0, 2]] = data[..., :-1, 1:3]            # root XZ velocity, one frame late (:374)
    pos = qrot(qinv(quat), pos)
    pos = torch.cumsum(pos, dim=-2)
    pos[..., 1] = data[..., 3]                            # root height (:380)
    return quat, pos


def recover_from_ric(data, joints_num=22):
    quat, r_pos = recover_root_rot_pos(data)
    positions = data[..., 4:(joints_num - 1) * 3 + 4]
    positions = positions.reshape(positions.shape[:-1] + (-1, 3))
    positions = qrot(qinv(quat[..., None, :]).expand(positions.shape[:-1] + (4,)), positions)
    positions[..., 0] += r_pos[..., 0:1]
    positions[..., 2] += r_pos[..., 2:3]
    return torch.cat([r_pos.unsqueeze(-2), positions], dim=-2)


def motion_temporal_filter(joints: np.ndarray, sigma: float = 1.0) -> np.ndarray:
    m = joints.reshape(joints.shape[0], -1).copy()
    for i in range(m.shape[1]):
        m[:, i] = gaussian_filter(m[:, i], sigma=sigma, mode="nearest")
    return m.reshape(m.shape[0], -1, 3)


def motion_to_joints(motion: torch.Tensor, mean, std, joints_num=22, sigma=1.0) -> np.ndarray:
    """One generated sample (len, 263), normalised -> filtered joints (len, 22, 3)   (tools/visualization.py:21-27,89)."""
    data = motion.numpy() * np.asarray(std) + np.asarray(mean)
    joint = recover_from_ric(torch.from_numpy(data).float(), joints_num).numpy()
    return motion_temporal_filter(joint, sigma) if sigma and sigma > 0 else joint
