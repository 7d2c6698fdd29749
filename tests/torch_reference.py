"""TEST INFRASTRUCTURE: an independent float64 restatement of the BEVLocalizer forward
(``snap/models/bev_localizer.py:130-220`` and everything below it) on torch's OWN primitives --
``F.conv2d``, ``F.max_pool2d``, ``F.interpolate(bilinear, align_corners=False)``,
``F.grid_sample(align_corners=False, padding_mode='border')`` for both the camera-ray lift and the
pose scoring, ``torch.sort`` for the view selection -- written against the reference's source, not
against ``oracle/``.  ``tests/test_oracle_composite_pin.py`` runs it next to ``oracle/model.py`` on
one small scene and demands agreement to 1e-9: a composite pin of the oracle by an implementation
the oracle did not generate (no numpy ``as_strided`` convolutions, no hand-rolled gathers).

Conventions: images / features channels-last on the outside (as the reference), NCHW inside the
torch primitives.  Nothing here is imported by the product.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

F64 = torch.float64


def t64(a):
  """float64 tensor; a torch tensor passes through ``.to`` (differentiable: the whole-model gradient
  test hands in float64 leaves that require grad)."""
  if isinstance(a, torch.Tensor):
    return a.to(F64)
  return torch.as_tensor(np.asarray(a), dtype=F64)


# ------------------------------------------------------------------------------------------
# encoder (snap/models/resnet.py, image_encoder.py, layers.py)
# ------------------------------------------------------------------------------------------
def _conv(x, kernel, stride=1, padding=(0, 0)):
  """flax.linen.Conv: NHWC activations, HWIO kernel (cross-correlation)."""
  w = t64(kernel).permute(3, 2, 0, 1)
  y = F.conv2d(x.permute(0, 3, 1, 2), w, stride=stride, padding=padding)
  return y.permute(0, 2, 3, 1)


def _std_kernel(kernel):
  """StdConv (resnet.py:73-79): per output channel over (H, W, I), eps 1e-10."""
  k = t64(kernel)
  var, mean = torch.var_mean(k, dim=(0, 1, 2), unbiased=False, keepdim=True)
  return (k - mean) / torch.sqrt(var + 1e-10)


def _group_norm(x, p, groups=32):
  """resnet.py:46-70 by hand (no F.group_norm): statistics over (H, W, channels of the group)."""
  n, h, w, c = x.shape
  g = x.reshape(n, h * w, groups, c // groups)
  var, mean = torch.var_mean(g, dim=(1, 3), unbiased=False, keepdim=True)
  g = (g - mean) / torch.sqrt(var + 1e-5)
  return g.reshape(n, h, w, c) * t64(p['scale']).reshape(1, 1, 1, c) + t64(p['bias']).reshape(1, 1, 1, c)


def _unit(p, x, stride, nmid):
  """Pre-activation bottleneck (resnet.py:103-132)."""
  nout = 4 * nmid
  residual = x
  y = torch.relu(_group_norm(x, p['gn1']))
  if x.shape[-1] != nout or stride != 1:
    residual = _conv(y, _std_kernel(p['conv_proj']['kernel']), stride)
  y = _conv(y, _std_kernel(p['conv1']['kernel']))
  y = torch.relu(_group_norm(y, p['gn2']))
  y = _conv(y, _std_kernel(p['conv2']['kernel']), stride, (1, 1))
  y = torch.relu(_group_norm(y, p['gn3']))
  y = _conv(y, _std_kernel(p['conv3']['kernel']))
  return y + residual


_BLOCKS = {26: [2, 2, 2, 2], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}


def _resnet(p, cfg, image):
  blocks = _BLOCKS.get(cfg['depth'], cfg['depth']) if not isinstance(cfg['depth'], (list, tuple)) else list(cfg['depth'])
  if cfg.get('limit_num_blocks') is not None:
    blocks = blocks[: cfg['limit_num_blocks']]
  width = int(64 * cfg['width'])
  x = image * 2 - 1
  if cfg['skip_root_block']:
    x = _conv(x, _std_kernel(p['conv_root']['kernel']), 1, (1, 1))
  else:
    x = _conv(x, _std_kernel(p['root_block']['conv_root']['kernel']), 2, (3, 3))
    x = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)     # pads with -inf
  out = []
  for i, size in enumerate(blocks):
    nmid = width * 2 ** i
    for u in range(size):
      x = _unit(p[f'block{i + 1}'][f'unit{u + 1:02d}'], x, 2 if (u == 0 and i > 0) else 1, nmid)
    out.append(x)
  return out


def image_encoder(p, cfg, image):
  """image_encoder.py:97-144 -> (finest FPN level cropped to the input, its stride (h, w))."""
  enc = cfg['encoder']
  nlev = len(enc['depth']) if isinstance(enc['depth'], (list, tuple)) else 4
  if enc.get('limit_num_blocks') is not None:
    nlev = min(nlev, enc['limit_num_blocks'])
  npyr = cfg.get('num_pyr_levels') or nlev
  max_stride = (0 if enc['skip_root_block'] else 2) + npyr - 1
  m = 2 ** max_stride
  h, w = image.shape[1:3]
  ph, pw = m - h % m, m - w % m                    # (a divisible size is padded by a full stride)
  padded = F.pad(image, (0, 0, 0, pw, 0, ph))
  stages = _resnet(p['encoder'], enc, padded)[:npyr][::-1]        # coarse -> fine
  f_prev = None
  for level, skip in enumerate(stages):
    dp = p['decoder']
    f = _group_norm(torch.relu(skip), dp[f'{level}_skip_norm'])
    f = _conv(f, dp[f'{level}_skip_conv']['kernel'])
    if f_prev is not None:
      up = F.interpolate(f_prev.permute(0, 3, 1, 2), scale_factor=2, mode='bilinear', align_corners=False)
      f = f + up.permute(0, 2, 3, 1)
    f_prev = f
  stride = (padded.shape[1] / f_prev.shape[1], padded.shape[2] / f_prev.shape[2])
  ch, cw = math.ceil(h / stride[0]), math.ceil(w / stride[1])
  return f_prev[:, :ch, :cw], stride


def _mlp(p, cfg, x):
  for i in range(len(cfg['layers'])):
    if i > 0 or cfg['apply_input_activation']:
      x = torch.relu(x)
    x = x @ t64(p[f'Dense_{i}']['kernel']) + t64(p[f'Dense_{i}']['bias'])
  return x


# ------------------------------------------------------------------------------------------
# camera-ray lift (snap/models/streetview_encoder.py, snap/utils/geometry.py)
# ------------------------------------------------------------------------------------------
def _project(cam, R, t, pts):
  """FisheyeCamera.world2image of scene points: pts [N, 3] -> (ij [N, 2], visible, depth)."""
  pv = (pts - t) @ R                                 # R^T (p - t)
  z = pv[:, 2]
  vis = z >= 1e-3
  xy = pv[:, :2] / torch.clamp(z, min=1e-3)[:, None]
  r2 = (xy ** 2).sum(-1)
  centre = r2 < 1e-6
  r = torch.sqrt(torch.where(centre, torch.full_like(r2, 1e-6), r2))
  th = torch.atan(r)
  k = cam['k_radial']
  dist = (1 + k[0] * th ** 2 + k[1] * th ** 4 + k[2] * th ** 6) * th / r
  dist = torch.where(centre, torch.ones_like(dist), dist)
  ok = centre | ((r < torch.tan(0.5 * cam['max_fov'])) & (dist > 0))
  p2d = xy * dist[:, None] * cam['f'] + cam['c']
  inside = ((p2d >= 0) & (p2d < cam['wh'])).all(-1)
  return p2d.flip(-1), vis & ok & inside, z


def _sample(fmap, ij):
  """interpolate_nd on an [h, w, D] map at (i, j) points: order-1 map_coordinates with the
  'nearest' extension == grid_sample(bilinear, border, align_corners=False)."""
  h, w, _ = fmap.shape
  grid = torch.stack([2 * ij[:, 1] / w - 1, 2 * ij[:, 0] / h - 1], -1)[None, None]     # (x, y)
  out = F.grid_sample(fmap.permute(2, 0, 1)[None], grid, mode='bilinear', padding_mode='border',
                      align_corners=False)
  return out[0, :, 0].T                                # [N, D]


def streetview_encoder(p, cfg, scene, xyz):
  """streetview_encoder.py:217-287 for ONE scene (weighted fusion).  Returns
  (image features [V, h, w, D], volume [X, Y, Z, D], valid [X, Y, Z])."""
  images = t64(scene['images'])
  V = images.shape[0]
  f_img, stride = image_encoder(p['image_encoder'], cfg['image_encoder'], images)
  proj = dict(layers=(cfg['feature_dim'] + cfg['num_scale_bins'],),
              apply_input_activation=cfg['proj_mlp']['apply_input_activation'])
  f_all = _mlp(p['proj_mlp'], proj, f_img)
  cam, T = scene['camera'], scene['T_view2scene']
  s = torch.tensor([1 / stride[1], 1 / stride[0]], dtype=T_dtype())      # (x, y) scale
  pts = xyz.reshape(-1, 3)
  fd, nb = cfg['feature_dim'], cfg['num_scale_bins']
  feats, vis, depth, dists = [], [], [], []
  for v in range(V):
    c = dict(wh=t64(cam.wh[v]) * s, f=t64(cam.f[v]) * s, c=t64(cam.c[v]) * s,
             k_radial=t64(cam.k_radial[v]), max_fov=t64(cam.max_fov[v]))
    ij, ok, z = _project(c, t64(T.R[v]), t64(T.t[v]), pts)
    feats.append(_sample(f_all[v], ij))
    vis.append(ok)
    depth.append(z)
    dists.append(torch.linalg.norm(pts - t64(T.t[v]), dim=-1))
  feats, vis, depth = torch.stack(feats, 1), torch.stack(vis, 1), torch.stack(depth, 1)   # [N, V, ...]
  k = cfg['top_k_view_selection']
  min_dist = None
  if k and V > k:                                    # jax.lax.top_k(-dist): k nearest VISIBLE views
    d = torch.where(vis, torch.stack(dists, 1), torch.full_like(depth, float('inf')))
    min_dist = d.min(-1).values
    sel = torch.sort(d, dim=-1, stable=True).indices[:, :k]
    feats = torch.gather(feats, 1, sel[..., None].expand(-1, -1, feats.shape[-1]))
    vis, depth = torch.gather(vis, 1, sel), torch.gather(depth, 1, sel)
  f_proj, bins = feats[..., :fd], feats[..., fd:]
  lo_d, hi_d = cfg['depth_min_max']
  tt = torch.log(torch.clamp(depth, lo_d, hi_d) / lo_d) / math.log(hi_d / lo_d)
  c = tt * (nb - 1)                                  # (index - 0.5 with index = 0.5 + t (S - 1))
  fl = torch.floor(c)
  wh = c - fl
  i0 = torch.clamp(fl.long(), 0, nb - 1)
  i1 = torch.clamp(fl.long() + 1, 0, nb - 1)
  score = (1 - wh) * torch.gather(bins, -1, i0[..., None])[..., 0] + wh * torch.gather(bins, -1, i1[..., None])[..., 0]
  # pool_multiview_features (:141-178): softmax(where=valid, initial=0) weights
  any_v = vis.any(-1)
  v_ = torch.where(any_v[:, None], vis, torch.ones_like(vis))
  sm = torch.where(v_, score, torch.full_like(score, -float('inf')))
  e = torch.where(v_, torch.exp(sm - torch.clamp(sm.max(-1, keepdim=True).values, min=0)), torch.zeros_like(sm))
  wgt = e / e.sum(-1, keepdim=True)
  mean = (wgt[..., None] * f_proj).sum(1)
  var = (wgt[..., None] * (f_proj - mean[:, None]) ** 2).sum(1)
  pooled = torch.cat([mean, var, sm.max(-1, keepdim=True).values], -1)
  pooled = torch.where(any_v[:, None], pooled, torch.zeros_like(pooled))
  valid = any_v
  if cfg.get('max_view_distance') is not None and min_dist is not None:
    valid = valid & (min_dist <= cfg['max_view_distance'])
  vol = _mlp(p['fusion_mlp'], cfg['fusion'], pooled)
  vol = torch.where(valid[:, None], vol, torch.zeros_like(vol))
  return f_img, vol.reshape(*xyz.shape[:-1], -1), valid.reshape(xyz.shape[:-1])


def T_dtype():
  return F64


# ------------------------------------------------------------------------------------------
# BEV mapper + localiser (bev_mapper.py, bev_localizer.py, pose_estimation.py)
# ------------------------------------------------------------------------------------------
def _max_pool_masked(feats, valid):
  """VerticalPooling('max') (bev_mapper.py:78-88) over axis -2 / modality fusion over stacked planes."""
  any_v = valid.any(-1)
  w = torch.where(any_v[..., None], valid, torch.ones_like(valid))[..., None]
  out = torch.where(w, feats, torch.full_like(feats, -float('inf'))).amax(-2)
  return torch.where(any_v[..., None], out, torch.zeros_like(out)), any_v


def bev_mapper(p, cfg, grid_extent, cell, scene, xy_bev=None):
  """One scene -> dict(image_features, volume, volume_valid, sv_plane, aerial_plane, matching, valid)."""
  T = scene['T_view2scene']
  if xy_bev is None:
    X, Y = grid_extent
    ii, jj = torch.meshgrid(torch.arange(X, dtype=T_dtype()), torch.arange(Y, dtype=T_dtype()), indexing='ij')
    xy_bev = (torch.stack([ii, jj], -1) + 0.5) * cell
  z0 = float(np.median(np.asarray(T.t)[..., -1])) - cfg.get('scene_z_offset', 4.0)
  z = torch.arange(0, cfg.get('scene_z_height', 12.0), cell, dtype=T_dtype()) + z0 + cell / 2
  xyz = torch.cat([xy_bev[:, :, None, :].expand(-1, -1, len(z), -1),
                   z[None, None, :, None].expand(*xy_bev.shape[:2], -1, -1)], -1)
  out = {}
  f_img, vol, vvalid = streetview_encoder(p['streetview_encoder'], cfg['streetview_encoder'], scene, xyz)
  out.update(image_features=f_img, volume=vol, volume_valid=vvalid)
  plane, pvalid = _max_pool_masked(vol, vvalid)
  out.update(sv_plane=plane, sv_plane_valid=pvalid)
  planes, valids = [plane], [pvalid]
  if cfg.get('aerial_encoder') is not None and 'rasters' in scene:
    fa, _ = image_encoder(p['aerial_encoder'], cfg['aerial_encoder'], t64(scene['rasters']['rgb'])[None])
    out['aerial_plane'] = fa[0]
    planes.append(fa[0])
    valids.append(torch.ones(fa.shape[1:3], dtype=torch.bool))
  if len(planes) > 1:
    assert cfg['modality_fusion']['pooling'] == 'max'
    plane, pvalid = _max_pool_masked(torch.stack(planes, -2), torch.stack(valids, -1))
  f = plane @ t64(p['matching_proj']['kernel']) + t64(p['matching_proj']['bias'])
  if cfg['normalize_matching_features']:             # layers.normalize (layers.py:45-52), eps 1e-5
    n = torch.linalg.norm(f, dim=-1, keepdim=True)
    f = torch.where(n < 1e-5, torch.zeros_like(f), f / torch.where(n < 1e-5, torch.ones_like(n), n))
  out['matching'] = torch.where(pvalid[..., None], f, torch.zeros_like(f))
  out['valid'] = pvalid
  return out


def bev_localizer(params, cfg, hfov_deg, grid_extent, cell, batch, pose_angles, pose_ts):
  """bev_localizer.py:130-220 on a batch of oracle-struct scenes; the RANSAC samples are inputs
  (pose_angles [B, P], pose_ts [B, P, 2]); the ground-truth pose is prepended as in the reference."""
  depth = cfg['query_frustum_depth']
  width = 3 * depth // 2
  W, D = int(round(width / cell)), int(round(depth / cell))
  ii, jj = torch.meshgrid(torch.arange(W, dtype=T_dtype()), torch.arange(D, dtype=T_dtype()), indexing='ij')
  q = (torch.stack([ii, jj], -1) + 0.5) * cell - torch.tensor([width / 2, 0.0], dtype=T_dtype())
  if cfg['filter_points_in_fov']:
    ang = torch.atan2(q[..., 0], q[..., 1])
    q = q[ang.abs() < math.radians(hfov_deg / 2)][:, None]          # [Nq, 1, 2]
  B = len(batch['query']['images'])
  out = {'map': [], 'query': [], 'sim': [], 'scores': []}
  pm, pq = params['bev_mapper'], params.get('bev_mapper_query', params['bev_mapper'])
  for b in range(B):
    sm = {k: (v[b] if not isinstance(v, dict) else {kk: vv[b] for kk, vv in v.items()})
          for k, v in batch['map'].items()}
    sq = {k: (v[b] if not isinstance(v, dict) else {kk: vv[b] for kk, vv in v.items()})
          for k, v in batch['query'].items()}
    m = bev_mapper(pm, cfg['bev_mapper'], grid_extent, cell, sm)
    qq = bev_mapper(pq, cfg.get('bev_mapper_query') or cfg['bev_mapper'], grid_extent, cell, sq, xy_bev=q)
    out['map'].append(m)
    out['query'].append(qq)
    fq = qq['matching'].reshape(-1, qq['matching'].shape[-1])
    vq = qq['valid'].reshape(-1)
    sim = torch.einsum('nd,ijd->nij', fq, m['matching'])
    if cfg['clip_negative_scores']:
      sim = torch.clamp(sim, min=0)
    if cfg['add_temperature']:
      sim = sim * torch.exp(t64(params['temperature']))
    sim = sim / max(int(vq.sum()), 1)
    out['sim'].append(sim)
    # poses: ground truth first (Transform2D.from_Transform3D: angle = atan2(R10, R00), t = t[:2])
    gt = batch['T_query2map']
    a0 = math.atan2(float(gt.R[b][1, 0]), float(gt.R[b][0, 0]))
    ang = torch.cat([torch.tensor([a0], dtype=T_dtype()), t64(pose_angles[b])])
    ts = torch.cat([t64(gt.t[b][:2])[None], t64(pose_ts[b])])
    qxy = q.reshape(-1, 2)
    cs, sn = torch.cos(ang), torch.sin(ang)
    uv = torch.stack([cs[:, None] * qxy[:, 0] - sn[:, None] * qxy[:, 1] + ts[:, 0:1],
                      sn[:, None] * qxy[:, 0] + cs[:, None] * qxy[:, 1] + ts[:, 1:2]], -1) / cell   # [P, Nq, 2]
    X, Y = sim.shape[1:]
    grid = torch.stack([2 * uv[..., 1] / Y - 1, 2 * uv[..., 0] / X - 1], -1)                  # (x, y)
    vals = F.grid_sample(sim[:, None], grid.permute(1, 0, 2)[:, :, None], mode='bilinear',
                         padding_mode='border', align_corners=False)[:, 0, :, 0]             # [Nq, P]
    assert not cfg['mask_score_out_of_bounds']
    out['scores'].append((vals * vq[:, None].to(T_dtype())).sum(0))
  return out
