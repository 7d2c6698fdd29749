"""Batched rigid transforms and camera models as torch tensor structs.

Mirrors the public surface of ``snap/utils/geometry.py`` (``Transform2D/3D`` with
``@``, ``.inv``, ``.magnitude()``; ``Camera`` / ``FisheyeCamera`` with ``.scale``
and ``.world2image``).  The reference builds these on ``dataclass_array``; here a
struct is a plain object whose fields share the leading batch shape and support
indexing.  These are host-side (tiny) tensors: the per-voxel projection of the
hot path runs in lift.hip, which consumes ``packed()`` views of these structs.
"""
import math

import torch


class _Struct:
  """Fields share a leading batch shape; indexing / mapping applies to all."""

  _fields = ()
  _event_ndim = {}

  def _map(self, fn):
    return type(self)(**{k: fn(getattr(self, k)) for k in self._fields})

  def __getitem__(self, idx):
    if not isinstance(idx, tuple):
      idx = (idx,)
    def index(k):
      v = getattr(self, k)
      nd = self._event_ndim[k]
      if any(i is Ellipsis for i in idx):
        return v[idx + (slice(None),) * nd]
      return v[idx]
    return type(self)(**{k: index(k) for k in self._fields})

  def to(self, *args, **kwargs):
    return self._map(lambda t: t.to(*args, **kwargs))

  def unsqueeze(self, dim):
    """Insert a batch axis (``dim`` counted over the batch shape, >= 0 or -1)."""
    def fn_for(k):
      nd = self._event_ndim[k]
      v = getattr(self, k)
      d = dim if dim >= 0 else v.dim() - nd + 1 + dim
      return v.unsqueeze(d)
    return type(self)(**{k: fn_for(k) for k in self._fields})

  @classmethod
  def cat(cls, items, dim=0):
    return cls(**{
        k: torch.cat([getattr(i, k) for i in items], dim) for k in cls._fields
    })

  def tree_flatten(self):
    return [getattr(self, k) for k in self._fields]

  def __repr__(self):
    return f'{type(self).__name__}(shape={tuple(self.shape)})'


class Transform3D(_Struct):
  """SE(3): ``R`` [..., 3, 3], ``t`` [..., 3]  (geometry.py:36-84)."""

  _fields = ('R', 't')
  _event_ndim = {'R': 2, 't': 1}

  def __init__(self, R, t):
    self.R = R
    self.t = t

  @classmethod
  def from_Rt(cls, R, t):
    return cls(R=R, t=t)

  @property
  def shape(self):
    return self.t.shape[:-1]

  @property
  def inv(self):
    R_inv = self.R.transpose(-1, -2)
    t_inv = -torch.einsum('...ij,...j->...i', R_inv, self.t)
    return Transform3D(R_inv, t_inv)

  def magnitude(self):
    trace = torch.diagonal(self.R, dim1=-2, dim2=-1).sum(-1)
    cos = torch.clamp((trace - 1) / 2, -1, 1)
    dr = torch.rad2deg(torch.abs(torch.arccos(cos)))
    dt = torch.linalg.norm(self.t, dim=-1)
    return dr, dt

  def transform(self, p3d):
    p3d = torch.einsum('...ij,...nj->...ni', self.R, p3d)
    return self.t[..., None, :] + p3d

  def compose(self, other):
    R = self.R @ other.R
    t = self.t + torch.einsum('...ij,...j->...i', self.R, other.t)
    return Transform3D(R, t)

  def __matmul__(self, other):
    if isinstance(other, Transform3D):
      return self.compose(other)
    if isinstance(other, torch.Tensor):
      return self.transform(other)
    raise TypeError(f'Unexpected type: {type(other)}')

  def packed(self):
    """[..., 12] = R row-major (9) | t (3): the layout lift.hip consumes."""
    return torch.cat([self.R.reshape(*self.shape, 9), self.t], -1).contiguous()


class Transform2D(_Struct):
  """SE(2): ``angle`` [...] (radians), ``t`` [..., 2]  (geometry.py:87-154)."""

  _fields = ('angle', 't')
  _event_ndim = {'angle': 0, 't': 1}

  def __init__(self, angle, t):
    self.angle = angle
    self.t = t

  @classmethod
  def from_radians(cls, angle, t):
    return cls(angle=angle, t=t)

  @classmethod
  def from_R(cls, R, t):
    angle = torch.atan2(R[..., 1, 0], R[..., 0, 0])
    return cls(angle, t)

  @classmethod
  def from_Transform3D(cls, transform):
    return cls.from_R(transform.R, transform.t[..., :2])

  @classmethod
  def from_packed(cls, p):
    """[..., 3] = (angle, tx, ty): the layout pose.hip produces / consumes."""
    return cls(p[..., 0], p[..., 1:])

  def packed(self):
    return torch.cat([self.angle[..., None], self.t], -1).contiguous()

  @property
  def shape(self):
    return self.angle.shape

  @property
  def R(self):
    cos, sin = torch.cos(self.angle), torch.sin(self.angle)
    return torch.stack([cos, -sin, sin, cos], -1).reshape(*self.shape, 2, 2)

  def _rotate(self, v, inverse=False, points=False):
    """R v (or R^T v) written out: cos x - sin y, sin x + cos y -- elementwise launches instead of
    the batched 2 x 2 matmul an einsum lowers to (40 004 two-by-two products per train step took
    0.3 ms as a bmm)."""
    cos, sin = torch.cos(self.angle), torch.sin(self.angle)
    if inverse:
      sin = -sin
    if points:
      cos, sin = cos[..., None], sin[..., None]
    x, y = v[..., 0], v[..., 1]
    return torch.stack([cos * x - sin * y, sin * x + cos * y], -1)

  @property
  def inv(self):
    return Transform2D(-self.angle, -self._rotate(self.t, inverse=True))

  def magnitude(self):
    dr = torch.rad2deg(torch.abs(self.angle)) % 360
    dr = torch.minimum(dr, 360 - dr)
    dt = torch.linalg.norm(self.t, dim=-1)
    return dr, dt

  def transform(self, points):
    return self.t[..., None, :] + self._rotate(points, points=True)

  def compose(self, other):
    angle = self.angle + other.angle
    t = self.t + self._rotate(other.t)
    return Transform2D(angle, t)

  def __matmul__(self, other):
    if isinstance(other, Transform2D):
      return self.compose(other)
    if isinstance(other, torch.Tensor):
      return self.transform(other)
    raise TypeError(f'Unexpected type: {type(other)}')


class Camera(_Struct):
  """Pinhole camera, half-integer pixel centres (geometry.py:160-221)."""

  _fields = ('wh', 'f', 'c')
  _event_ndim = {'wh': 1, 'f': 1, 'c': 1}
  eps = 1e-3
  is_fisheye = False

  def __init__(self, wh, f, c):
    self.wh = wh
    self.f = f
    self.c = c

  @property
  def shape(self):
    return self.wh.shape[:-1]

  def scale(self, scale):
    return Camera(self.wh * scale, self.f * scale, self.c * scale)

  def in_image(self, p2d):
    return torch.all((p2d >= 0) & (p2d < self.wh[..., None, :]), -1)

  def project(self, p3d):
    z = p3d[..., -1]
    valid = z >= self.eps
    z = z.clamp(min=self.eps)[..., None]
    return p3d[..., :-1] / z, valid

  def denormalize(self, p2d):
    return p2d * self.f[..., None, :] + self.c[..., None, :]

  def normalize(self, p2d):
    return (p2d - self.c[..., None, :]) / self.f[..., None, :]

  def world2image(self, p3d):
    p2d, visible = self.project(p3d)
    p2d = self.denormalize(p2d)
    return p2d, visible & self.in_image(p2d)

  def packed(self):
    """[..., 11] = wh f c k_radial(0) max_fov(pi) pad."""
    z = torch.zeros(*self.shape, 3, dtype=self.wh.dtype, device=self.wh.device)
    fov = torch.full((*self.shape, 1), math.pi, dtype=self.wh.dtype, device=self.wh.device)
    pad = torch.zeros_like(fov)
    return torch.cat([self.wh, self.f, self.c, z, fov, pad], -1).contiguous()


class FisheyeCamera(Camera):
  """Equidistant fisheye with polynomial radial distortion (geometry.py:224-280)."""

  _fields = ('wh', 'f', 'c', 'k_radial', 'max_fov')
  _event_ndim = {'wh': 1, 'f': 1, 'c': 1, 'k_radial': 1, 'max_fov': 0}
  is_fisheye = True

  def __init__(self, wh, f, c, k_radial, max_fov):
    super().__init__(wh, f, c)
    self.k_radial = k_radial
    self.max_fov = max_fov

  @classmethod
  def from_dict(cls, intrinsics):
    K = intrinsics['K']
    wh = torch.stack(
        [torch.as_tensor(intrinsics['image_width']),
         torch.as_tensor(intrinsics['image_height'])], -1
    ).to(K.dtype)
    f = torch.stack([K[..., 0, 0], K[..., 1, 1]], -1)
    c = torch.stack([K[..., 0, 2], K[..., 1, 2]], -1)
    k_radial = intrinsics['distortion']['radial']
    max_fov = intrinsics.get('maxfov')
    if max_fov is None:
      max_fov = torch.full(wh.shape[:-1], math.radians(115.0), dtype=K.dtype)
    return cls(wh, f, c, k_radial, max_fov)

  def scale(self, scale):
    return FisheyeCamera(
        self.wh * scale, self.f * scale, self.c * scale, self.k_radial,
        self.max_fov,
    )

  def distort_points(self, p2d):
    radius2 = torch.sum(p2d**2, dim=-1)
    in_center = radius2 < self.eps**2
    radius = torch.sqrt(torch.where(in_center, torch.full_like(radius2, self.eps**2), radius2))
    theta = torch.arctan(radius)
    theta2 = theta**2
    k = self.k_radial[..., None, :]
    offset = sum(k[..., i] * theta2 ** (i + 1) for i in range(3))
    dist = (offset + 1) * theta / radius
    dist = torch.where(in_center, torch.ones_like(dist), dist)
    valid = in_center | (
        (radius < torch.tan(0.5 * self.max_fov[..., None])) & (dist > 0)
    )
    return p2d * dist[..., None], valid

  def world2image(self, p3d):
    p2d, visible = self.project(p3d)
    p2d, valid = self.distort_points(p2d)
    p2d = self.denormalize(p2d)
    return p2d, visible & valid & self.in_image(p2d)

  def packed(self):
    """[..., 11] = wh f c k_radial max_fov tan(max_fov / 2): the last entry saves the lift
    kernels a tanf per (voxel, view)."""
    tan_half = torch.tan(0.5 * self.max_fov)[..., None]
    return torch.cat(
        [self.wh, self.f, self.c, self.k_radial, self.max_fov[..., None], tan_half], -1
    ).contiguous()
