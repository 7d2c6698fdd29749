"""Oracle (test infrastructure): rigid transforms and camera models in numpy.

Restates ``snap/utils/geometry.py`` (reference file:line cited per method).
Batched structs: every field shares the leading batch shape.
"""
import numpy as np


class Transform3D:
  """SE(3) transform (snap/utils/geometry.py:36-84)."""

  def __init__(self, R, t):
    self.R = np.asarray(R)
    self.t = np.asarray(t)

  @property
  def shape(self):
    return self.t.shape[:-1]

  def __getitem__(self, idx):
    return Transform3D(self.R[idx], self.t[idx])

  @property
  def inv(self):
    # geometry.py:52-56
    R_inv = np.swapaxes(self.R, -1, -2)
    t_inv = -np.einsum('...ij,...j->...i', R_inv, self.t)
    return Transform3D(R_inv, t_inv)

  def magnitude(self):
    # geometry.py:58-65
    trace = np.trace(self.R, axis1=-2, axis2=-1)
    cos = np.clip((trace - 1) / 2, -1, 1)
    dr = np.rad2deg(np.abs(np.arccos(cos)))
    dt = np.linalg.norm(self.t, axis=-1)
    return dr, dt

  def transform(self, p3d):
    # geometry.py:67-69
    p3d = np.einsum('...ij,...nj->...ni', self.R, p3d)
    return self.t[..., None, :] + p3d

  def compose(self, other):
    # geometry.py:71-74
    R = self.R @ other.R
    t = self.t + np.einsum('...ij,...j->...i', self.R, other.t)
    return Transform3D(R, t)

  def __matmul__(self, other):
    if isinstance(other, Transform3D):
      return self.compose(other)
    return self.transform(np.asarray(other))


class Transform2D:
  """SE(2) transform (snap/utils/geometry.py:87-154)."""

  def __init__(self, angle, t):
    self.angle = np.asarray(angle)
    self.t = np.asarray(t)

  @classmethod
  def from_radians(cls, angle, t):
    return cls(angle, t)

  @classmethod
  def from_R(cls, R, t):
    # geometry.py:102-106
    angle = np.arctan2(R[..., 1, 0], R[..., 0, 0])
    return cls(angle, t)

  @classmethod
  def from_Transform3D(cls, transform):
    # geometry.py:108-110
    return cls.from_R(transform.R, transform.t[..., :2])

  @property
  def shape(self):
    return self.angle.shape

  def __getitem__(self, idx):
    return Transform2D(self.angle[idx], self.t[idx])

  @property
  def R(self):
    # geometry.py:112-117
    cos = np.cos(self.angle)
    sin = np.sin(self.angle)
    R_flat = np.stack([cos, -sin, sin, cos], -1)
    return R_flat.reshape(*self.shape, 2, 2)

  @property
  def inv(self):
    # geometry.py:125-129
    R_inv = np.swapaxes(self.R, -1, -2)
    t_inv = -np.einsum('...ij,...j->...i', R_inv, self.t)
    return Transform2D(-self.angle, t_inv)

  def magnitude(self):
    # geometry.py:131-135
    dr = np.rad2deg(np.abs(self.angle)) % 360
    dr = np.minimum(dr, 360 - dr)
    dt = np.linalg.norm(self.t, axis=-1)
    return dr, dt

  def transform(self, points):
    # geometry.py:137-139
    points = np.einsum('...ij,...nj->...ni', self.R, points)
    return self.t[..., None, :] + points

  def compose(self, other):
    # geometry.py:141-144
    angle = self.angle + other.angle
    t = self.t + np.einsum('...ij,...j->...i', self.R, other.t)
    return Transform2D(angle, t)

  def __matmul__(self, other):
    if isinstance(other, Transform2D):
      return self.compose(other)
    return self.transform(np.asarray(other))


class Camera:
  """Pinhole camera (snap/utils/geometry.py:160-221)."""

  eps = 1e-3

  def __init__(self, wh, f, c):
    self.wh = np.asarray(wh)
    self.f = np.asarray(f)
    self.c = np.asarray(c)

  @property
  def shape(self):
    return self.wh.shape[:-1]

  def __getitem__(self, idx):
    return Camera(self.wh[idx], self.f[idx], self.c[idx])

  def scale(self, scale):
    # geometry.py:179-183
    return Camera(self.wh * scale, self.f * scale, self.c * scale)

  def in_image(self, p2d):
    # geometry.py:193-196
    return np.all((p2d >= 0) & (p2d < self.wh[..., None, :]), -1)

  def project(self, p3d):
    # geometry.py:198-205
    z = p3d[..., -1]
    valid = z >= self.eps
    z = np.clip(z, self.eps, None)[..., None]
    p2d = p3d[..., :-1] / z
    return p2d, valid

  def denormalize(self, p2d):
    # geometry.py:207-210
    return p2d * self.f[..., None, :] + self.c[..., None, :]

  def world2image(self, p3d):
    # geometry.py:216-221
    p2d, visible = self.project(p3d)
    p2d = self.denormalize(p2d)
    valid = visible & self.in_image(p2d)
    return p2d, valid


class FisheyeCamera(Camera):
  """Equidistant-polynomial fisheye (snap/utils/geometry.py:224-280)."""

  def __init__(self, wh, f, c, k_radial, max_fov):
    super().__init__(wh, f, c)
    self.k_radial = np.asarray(k_radial)
    self.max_fov = np.asarray(max_fov)

  def __getitem__(self, idx):
    return FisheyeCamera(
        self.wh[idx], self.f[idx], self.c[idx], self.k_radial[idx],
        self.max_fov[idx],
    )

  def scale(self, scale):
    # geometry.py:250-258
    return FisheyeCamera(
        self.wh * scale, self.f * scale, self.c * scale, self.k_radial,
        self.max_fov,
    )

  def distort_points(self, p2d):
    # geometry.py:260-272.  p2d: [..., n, 2]; camera fields: [...].
    dtype = p2d.dtype
    eps2 = np.asarray(self.eps**2, dtype)
    radius2 = np.sum(p2d**2, axis=-1)
    in_center = radius2 < eps2
    radius = np.sqrt(np.where(in_center, eps2, radius2))
    theta = np.arctan(radius)
    theta2 = theta**2
    k = self.k_radial[..., None, :]  # broadcast over points
    offset = sum(k[..., i] * theta2 ** (i + 1) for i in range(3))
    dist = (offset + 1) * theta / radius
    dist = np.where(in_center, np.asarray(1.0, dtype), dist)
    p2d_dist = p2d * dist[..., None]
    max_fov = self.max_fov[..., None]
    valid = in_center | ((radius < np.tan(0.5 * max_fov)) & (dist > 0))
    return p2d_dist.astype(dtype), valid

  def world2image(self, p3d):
    # geometry.py:274-280
    p2d, visible = self.project(p3d)
    p2d, valid = self.distort_points(p2d)
    p2d = self.denormalize(p2d)
    valid = visible & valid & self.in_image(p2d)
    return p2d, valid
