"""Synthetic batches with the schema of ``snap/data/loader.py:82-168``.

The reference's dataset is not released (README.md:32); benchmarks, smoke tests and
parity tests use seeded synthetic scenes: images and aerial
tiles ~U[0,1), equidistant fisheye cameras with f = 0.6 W, c = wh/2, k_radial = 0,
max_fov = 115 deg, map cameras inside the central half of the grid at ~2 m height
with random yaw, and a planted ``T_query2map``.
"""
import math

import numpy as np
import torch

from snap_amd.data import types as data_types
from snap_amd.utils import geometry
from snap_amd.utils import grids


def _yaw_camera_rotation(yaw):
  """R (view -> scene) for a horizontal optical axis at `yaw`: columns = right, down, forward."""
  c, s = np.cos(yaw), np.sin(yaw)
  right = np.stack([s, -c, np.zeros_like(c)], -1)
  down = np.stack([np.zeros_like(c), np.zeros_like(c), -np.ones_like(c)], -1)
  fwd = np.stack([c, s, np.zeros_like(c)], -1)
  return np.stack([right, down, fwd], -1)


def make_cameras(B, V, image_size, k_radial=0.0, max_fov_deg=115.0):
  H, W = image_size
  wh = np.broadcast_to(np.array([W, H], np.float32), (B, V, 2)).copy()
  f = np.full((B, V, 2), 0.6 * W, np.float32)
  c = wh / 2
  k = np.full((B, V, 3), k_radial, np.float32)
  fov = np.full((B, V), math.radians(max_fov_deg), np.float32)
  return wh, f, c, k, fov


SURFEL_ROAD_CLASSES = ('crosswalk', 'sidewalk', 'pavedroad', 'stopline', 'line', 'otherlanemarking')


def make_batch(
    batch_size, grid: grids.Grid3D, num_views, image_size, query_image_size=None,
    seed=0, device='cpu', with_aerial=True, with_gt=True, k_radial=0.0,
    semantic_classes=None,
):
  """Returns a batch dict: map / query scenes + planted T_query2map."""
  rng = np.random.default_rng(seed)
  B, V = batch_size, num_views
  H, W = image_size
  qH, qW = query_image_size or image_size
  ext = grid.extent_meters  # (x, y, z) metres
  X, Y = grid.extent[:2]

  def t(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(device)

  # map views: centres in the central half of the footprint at ~2 m.
  cx = rng.uniform(0.25 * ext[0], 0.75 * ext[0], (B, V))
  cy = rng.uniform(0.25 * ext[1], 0.75 * ext[1], (B, V))
  cz = rng.uniform(1.8, 2.2, (B, V))
  yaw = rng.uniform(0, 2 * np.pi, (B, V))
  R_map = _yaw_camera_rotation(yaw)
  t_map = np.stack([cx, cy, cz], -1)
  wh, f, c, k, fov = make_cameras(B, V, (H, W), k_radial)
  map_scene = {
      'images': t(rng.random((B, V, H, W, 3), dtype=np.float32)),
      'camera': geometry.FisheyeCamera(t(wh), t(f), t(c), t(k), t(fov)),
      'T_view2scene': geometry.Transform3D(t(R_map), t(t_map)),
  }
  if with_aerial:
    map_scene['rasters'] = {'rgb': t(rng.random((B, X, Y, 3), dtype=np.float32))}
  if semantic_classes:
    # boolean rasters [B, X, Y, N] (loader.py:150-158): random blobs per class; the surfel-road
    # classes are made mutually exclusive as in the data (semantic_raster_encoder.py:33-35)
    srng = np.random.default_rng(seed + 977)     # own stream: the other fields stay unchanged
    sem = srng.random((B, X, Y, len(semantic_classes))) < 0.3
    road = [i for i, c in enumerate(semantic_classes) if c in SURFEL_ROAD_CLASSES]
    if road:
      pick = srng.integers(0, len(road) + 1, (B, X, Y))
      for k, i in enumerate(road):
        sem[..., i] = pick == k
    map_scene.setdefault('rasters', {})['semantics'] = torch.as_tensor(sem).to(device)

  # query: one view at the origin of its gravity-aligned frame, looking along +y.
  R_q = _yaw_camera_rotation(np.full((B, 1), np.pi / 2))
  t_q = np.stack(
      [np.zeros((B, 1)), np.zeros((B, 1)), rng.uniform(1.8, 2.2, (B, 1))], -1
  )
  wh, f, c, k, fov = make_cameras(B, 1, (qH, qW), k_radial)
  query_scene = {
      'images': t(rng.random((B, 1, qH, qW, 3), dtype=np.float32)),
      'camera': geometry.FisheyeCamera(t(wh), t(f), t(c), t(k), t(fov)),
      'T_view2scene': geometry.Transform3D(t(R_q), t(t_q)),
  }
  batch = {
      'map': map_scene,
      'query': query_scene,
      'batch_mask': torch.ones(B, dtype=torch.bool, device=device),
  }
  if with_gt:
    theta = rng.uniform(0, 2 * np.pi, B)
    tx = rng.uniform(0.3 * ext[0], 0.7 * ext[0], B)
    ty = rng.uniform(0.3 * ext[1], 0.7 * ext[1], B)
    cth, sth = np.cos(theta), np.sin(theta)
    Rg = np.zeros((B, 3, 3))
    Rg[:, 0, 0], Rg[:, 0, 1], Rg[:, 1, 0], Rg[:, 1, 1], Rg[:, 2, 2] = cth, -sth, sth, cth, 1
    tg = np.stack([tx, ty, np.zeros(B)], -1)
    batch['T_query2map'] = geometry.Transform3D(t(Rg), t(tg))
  return batch


def meta_data(cell_size=0.2, grid_size=(24, 32, 12), hfov_deg=72.0):
  """Dataset meta data consumed by ``BEVLocalizerModel`` (loader.py:424-433)."""
  scene = data_types.SceneConfig(grid_size=tuple(grid_size), streetview_hfov_deg=hfov_deg)
  grid = grids.Grid3D.from_extent_meters(scene.grid_size, cell_size)
  return {
      'grid': grid,
      'build_config': data_types.BuildConfig(scene_config=scene),
      'semantic_map_classes': None,
  }
