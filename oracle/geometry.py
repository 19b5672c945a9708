"""Geometry CPU restatement (oracle; test infrastructure only).

Reference: /root/reference/src/utils/alignment.py:11-222,
/root/reference/src/utils/rotation_conversions.py:38-120, 418-571,
/root/reference/src/utils/quaternion.py:54-73 (qrot), :386-396 (qbetween), :28-30 (qnormalize).
Branch-free formulations (torch.where) of the reference's masked-index code; same values.
"""
import torch
import torch.nn.functional as F

FACE_JOINT_INDX = [2, 1, 17, 16]  # /root/reference/src/utils/paramUtil.py:89


def _f(t):
    """The reference's ``.float()`` casts (no-ops on its fp32 tensors).  A float64 tensor stays float64, so that the SAME restatement run on
    ``.double()`` weights and inputs is the high-precision yardstick of tests/parity_tol.py (what the fp32 arithmetic is an approximation of)."""
    return t if t.dtype == torch.float64 else t.float()


def rotation_6d_to_matrix(d6):
    """rotation_conversions.py:511-534 -- note the interleaved [0,2,4],[1,3,5] shuffle (:527-528)."""
    a1, a2 = d6[..., [0, 2, 4]], d6[..., [1, 3, 5]]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def matrix_to_rotation_6d(m):
    """rotation_conversions.py:540-571 (first two rows, interleaved)."""
    d6 = m[..., :2, :].reshape(*m.shape[:-2], 6)
    return d6[..., [0, 3, 1, 4, 2, 5]]


def _sqrt_positive_part(x):
    """rotation_conversions.py:85-95."""
    return torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0)), torch.zeros_like(x))


def _copysign(a, b):
    """rotation_conversions.py:68-82."""
    return torch.where((a < 0) != (b < 0), -a, a)


def matrix_to_quaternion(m):
    """rotation_conversions.py:98-120."""
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    o0 = 0.5 * _sqrt_positive_part(1 + m00 + m11 + m22)
    x = 0.5 * _sqrt_positive_part(1 + m00 - m11 - m22)
    y = 0.5 * _sqrt_positive_part(1 - m00 + m11 - m22)
    z = 0.5 * _sqrt_positive_part(1 - m00 - m11 + m22)
    o1 = _copysign(x, m[..., 2, 1] - m[..., 1, 2])
    o2 = _copysign(y, m[..., 0, 2] - m[..., 2, 0])
    o3 = _copysign(z, m[..., 1, 0] - m[..., 0, 1])
    return torch.stack((o0, o1, o2, o3), -1)


def _half_sinc(half_angles, angles):
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    return torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half_angles) / safe)


def quaternion_to_axis_angle(q):
    """rotation_conversions.py:480-508."""
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    return q[..., 1:] / _half_sinc(half, angles)


def axis_angle_to_quaternion(aa):
    """rotation_conversions.py:449-477."""
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * angles
    return torch.cat([torch.cos(half), aa * _half_sinc(half, angles)], dim=-1)


def quaternion_to_matrix(q):
    """rotation_conversions.py:38-65."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def qrot(q, v):
    """quaternion.py:54-73."""
    qvec = q[..., 1:]
    uv = torch.cross(qvec, v, dim=-1)
    uuv = torch.cross(qvec, uv, dim=-1)
    return v + 2 * (q[..., :1] * uv + uuv)


def qbetween(v0, v1):
    """quaternion.py:386-396 (+ qnormalize :28-30)."""
    v = torch.cross(v0, v1, dim=-1)
    w = torch.sqrt((v0 ** 2).sum(-1, keepdim=True) * (v1 ** 2).sum(-1, keepdim=True)) \
        + (v0 * v1).sum(-1, keepdim=True) + 1e-8
    q = torch.cat([w, v], dim=-1)
    return q / torch.norm(q, dim=-1, keepdim=True)


def ih_to_smpl(motion):
    """alignment.py:11-38: 262 -> 205 (rot6d -> -axis_angle, 6 zero pad, last 4 kept)."""
    B = motion.shape[0]
    poses = _f(motion[:, :, 132:132 + 126].reshape(B, -1, 21, 6))
    poses = quaternion_to_axis_angle(matrix_to_quaternion(rotation_6d_to_matrix(poses))) * -1
    poses = poses.reshape(B, -1, 63)
    pad = torch.zeros(B, poses.shape[1], 6, dtype=poses.dtype)
    return torch.cat([motion[:, :, :132], poses, pad, motion[:, :, -4:]], dim=2)


def smpl_to_ih(motion):
    """alignment.py:40-67: 205/201 -> 262; appends motion[..., -4:] of whatever came in."""
    B = motion.shape[0]
    poses = _f(motion[:, :, 132:132 + 69].reshape(B, -1, 23, 3)) * -1
    poses = matrix_to_rotation_6d(quaternion_to_matrix(axis_angle_to_quaternion(poses)))
    poses = poses.reshape(B, -1, 138)[:, :, :-12]
    return torch.cat([motion[:, :, :132], poses, motion[:, :, -4:]], dim=2)


def align_motions(motion1, motion2, diag=None):
    """alignment.py:112-158 + align_trajectories :69-109, mask=None.  Returns the moved motion2 (201-d).
    diag (tests only): a list that receives this call's conditioning figures per sample -- the XZ root displacement lengths of the two
    trajectories (the rotation is the angle between their DIRECTIONS: a near-stationary root makes it rounding noise), the real part of
    the un-normalised qbetween quaternion (-> 0 for anti-parallel directions) and the largest |position| the rotation is applied to."""
    B = motion1.shape[0]
    p1 = motion1[..., :66].reshape(B, -1, 22, 3)
    p2 = motion2[..., :66].reshape(B, -1, 22, 3)
    v2 = motion2[..., 66:132].reshape(B, -1, 22, 3)
    r2 = motion2[..., 132:132 + 69]
    p2 = p2 + (p1[:, 0, 0] - p2[:, 0, 0])[:, None, None, :]
    t1, t2 = p1[:, :, 0], p2[:, :, 0]
    d1 = (t1[:, -1] - t1[:, 0]).clone()
    d2 = (t2[:, -1] - t2[:, 0]).clone()
    d1[:, 1] = 0
    d2[:, 1] = 0
    if diag is not None:
        n1, n2 = d1.norm(dim=1), d2.norm(dim=1)
        u1, u2 = d1 / torch.sqrt((d1 ** 2).sum(dim=1, keepdim=True) + 1e-8), d2 / torch.sqrt((d2 ** 2).sum(dim=1, keepdim=True) + 1e-8)
        diag.append(dict(disp_target=n1, disp_moved=n2, w=1 + (u1 * u2).sum(dim=1), reach=(p2 - p2[:, :1, :1]).abs().amax(dim=(1, 2, 3))))
    d1 = d1 / torch.sqrt((d1 ** 2).sum(dim=1, keepdim=True) + 1e-8)
    d2 = d2 / torch.sqrt((d2 ** 2).sum(dim=1, keepdim=True) + 1e-8)
    q = qbetween(d2, d1)[:, None, None, :].expand(-1, p2.shape[1], 22, -1)
    p2 = qrot(q, p2)
    p2 = p2 + (p1[:, 0, 0] - p2[:, 0, 0])[:, None, None, :]
    v2 = qrot(q, v2)
    return torch.cat([p2.reshape(B, -1, 66), v2.reshape(B, -1, 66), r2], dim=-1)


def center_motion(motion, diag=None):
    """alignment.py:161-222 (201-d output).  torch.cross there has no dim (quirk 9): B != 3 assumed.
    diag (tests only): receives per sample the first-frame hip distance |across|, the length of Y x across before normalisation
    (-> 0 when the hips are stacked vertically) and the real part of the un-normalised facing quaternion (-> 0 when facing -Z)."""
    B = motion.shape[0]
    pos = motion[:, :, :66].reshape(B, -1, 22, 3).clone()
    vel = motion[:, :, 66:132].reshape(B, -1, 22, 3)
    rot = motion[:, :, 132:132 + 69]
    floor = pos.min(dim=1).values.min(dim=1).values[:, 1]
    pos[:, :, :, 1] -= floor[:, None, None]
    root_init = pos[:, 0]
    xz = root_init[:, 0] * torch.tensor([1.0, 0.0, 1.0], dtype=pos.dtype)
    pos2 = pos - xz[:, None, None, :]
    r_hip, l_hip = FACE_JOINT_INDX[:2]
    across = root_init[:, r_hip] - root_init[:, l_hip]
    across_len = torch.sqrt((across ** 2).sum(dim=-1))
    across = across / across_len.unsqueeze(-1)
    fwd = torch.cross(torch.tensor([0.0, 1.0, 0.0], dtype=pos.dtype).expand(B, -1), across, dim=-1)
    fwd_len = torch.sqrt((fwd ** 2).sum(dim=-1))
    fwd = fwd / fwd_len.unsqueeze(-1)
    if diag is not None:
        diag.append(dict(across=across_len, fwd=fwd_len, w=1 + fwd[:, 2], reach=pos2.abs().amax(dim=(1, 2, 3))))
    target = torch.tensor([0.0, 0.0, 1.0], dtype=pos.dtype).expand(B, -1)
    q = qbetween(fwd, target)[:, None, None, :].expand(-1, pos2.shape[1], 22, -1)
    pos2 = qrot(q, pos2)
    vel = qrot(q, vel)
    return torch.cat([pos2.reshape(B, -1, 66), vel.reshape(B, -1, 66), rot], dim=-1)
