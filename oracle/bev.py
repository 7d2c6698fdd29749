"""Oracle (test infrastructure): BEV mapper (vertical pooling, fusion, matching head).

Restates ``snap/models/bev_mapper.py``.
"""
import numpy as np

from oracle import encoder
from oracle import lift


def log_sigmoid(x):
  """jax.nn.log_sigmoid = -softplus(-x), evaluated stably."""
  return np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))


def masked_softmax(x, where, axis=-1):
  """jax.nn.softmax(x, where=where, initial=0, axis): max over the selected entries and the
  initial value 0; sum over the selected entries; unselected outputs are 0."""
  shift = np.maximum(np.max(np.where(where, x, -np.inf), axis=axis, keepdims=True), 0)
  e = np.where(where, np.exp(x - shift), 0)
  return e / e.sum(axis, keepdims=True)


def layers_masked_softmax(x, mask, axis=-1):
  """snap/models/layers.py:38-43: an all-false mask acts as all-true; masked entries -> -inf."""
  valid = mask.any(axis=axis, keepdims=True)
  mask = np.where(valid, mask, True)
  xm = np.where(mask, x, -np.inf)
  e = np.exp(xm - xm.max(axis=axis, keepdims=True))
  return e / e.sum(axis=axis, keepdims=True)


def vertical_pooling(config, features, valid, params=None):
  """bev_mapper.py:56-88.

  features [..., Z, D], valid [..., Z] -> plane features [..., D], valid [...]
  (+ 'scores' / 'weights' [..., Z] for the 'softmax' / 'weighted' modes).  ``params``:
  {'confidence_head': {'kernel' [D,1], 'bias' [1]}} or {'fusion_mlp': ...} ('mlp').
  """
  dtype = features.dtype
  valid_any = valid.any(-1)
  valid_any_or_all = np.where(valid_any[..., None], valid, True)
  where = valid_any_or_all[..., None]
  pooling = config['pooling']
  extra = {}
  if pooling in ('weighted', 'softmax'):
    head = params['confidence_head']
    scores = (features @ head['kernel'].astype(dtype))[..., 0] + head['bias'].astype(dtype)[0]
    if pooling == 'weighted':
      scores = log_sigmoid(scores)
    weights = masked_softmax(scores, valid_any_or_all, axis=-1)
    weights = np.where(valid, weights, 0).astype(dtype)
    out = (features * weights[..., None]).sum(-2)
    extra = dict(scores=scores.astype(dtype), weights=weights)
  elif pooling == 'mlp':
    from oracle import encoder as o_enc
    f = np.where(valid[..., None], features, 0)
    f = f.reshape(*f.shape[:-2], -1)
    out = o_enc.mlp(params['fusion_mlp'], config['mlp'], f)
  elif pooling == 'max':
    out = np.where(where, features, -np.inf).max(-2)
  elif pooling == 'sum':
    out = np.where(where, features, 0).sum(-2)
  elif pooling == 'mean':
    out = np.where(where, features, 0).sum(-2) / where.sum(-2)
  else:
    raise NotImplementedError(pooling)
  out = np.where(valid_any[..., None], out, 0).astype(dtype)
  return dict(features=out, valid=valid_any, **extra)


def build_xyz_query(config, grid, scene_t_view, xy_bev=None, z_offset=None):
  """bev_mapper.py:162-196 (eval path: no random z offset).

  Returns xyz_query [B, X, Y, Z, 3].
  """
  t = scene_t_view.t
  dtype = t.dtype
  B = t.shape[0]
  xy = xy_bev
  if xy is None:
    xy = grid.index_to_xyz(grid.grid_index(), dtype)
  if xy.ndim != 4:
    xy = np.repeat(xy[None], B, axis=0)
  if z_offset is None:
    camera_heights = np.median(t[..., -1], axis=-1)
    height_below_camera = config.get('scene_z_offset', 4.0)
    z_offset = (camera_heights - height_below_camera).astype(dtype)
  scene_z_height = config.get('scene_z_height', 12.0)
  cell = grid.cell_size
  z = (
      np.arange(0, scene_z_height, cell).astype(dtype)[None]
      + z_offset[:, None]
      + dtype.type(cell / 2)
  ).astype(dtype)
  xy_b, z_b = np.broadcast_arrays(
      xy[:, :, :, None, :], z[:, None, None, :, None]
  )
  return np.concatenate([xy_b, z_b[..., :1]], axis=-1).astype(dtype)


def fuse_neural_maps(config, planes, params=None):
  """bev_mapper.py:225-252 (modality dropout never fires: train is not forwarded)."""
  if len(planes) == 1:
    return planes[0]
  features = np.stack([p['features'] for p in planes], axis=-2)
  valid = np.stack([p['valid'] for p in planes], axis=-1)
  out = vertical_pooling(config['modality_fusion'], features, valid, params)
  return dict(features=out['features'], valid=out['valid'])


def matching_head(params, config, plane):
  """bev_mapper.py:284-291."""
  f = encoder.dense(params['matching_proj'], plane['features'])
  if config['normalize_matching_features']:
    f = encoder.normalize(f)
  f = np.where(plane['valid'][..., None], f, 0).astype(plane['features'].dtype)
  return dict(features=f, valid=plane['valid'])


def bev_mapper(params, config, grid, data):
  """bev_mapper.py:254-296 (streetview [+ aerial] [+ semantic] modalities, eval).  The raster
  class names of the semantic modality travel as ``config['_semantic_map_classes']``."""
  pred = {}
  planes = []
  data = dict(data)
  if config.get('streetview_encoder') is not None:
    if 'xyz_query' not in data:
      data['xyz_query'] = build_xyz_query(
          config, grid, data['T_view2scene'], data.get('xy_bev'),
          data.get('z_offset'),
      )
    sv = lift.streetview_encoder(
        params['streetview_encoder'], config['streetview_encoder'], data
    )
    vol = sv['feature_volume']
    sv['vertical_pooling'] = {}
    sv['feature_plane'] = vertical_pooling(
        config['pooling'], vol['features'], vol['valid']
    )
    pred['streetview'] = sv
    planes.append(sv['feature_plane'])
  if config.get('aerial_encoder') is not None and 'rasters' in data:
    pyr = encoder.image_encoder(
        params['aerial_encoder'], config['aerial_encoder'],
        data['rasters']['rgb'],
    )
    f = pyr['features'][-1]
    plane = dict(features=f, valid=np.ones(f.shape[:-1], bool))
    pred['aerial'] = {'feature_plane': plane}
    planes.append(plane)
  if config.get('semantic_encoder') is not None and 'rasters' in data:
    # bev_mapper.py:214-223,273-278 (no semantic rasters for query images)
    pyr = semantic_raster_encoder(
        params['semantic_encoder'], config['semantic_encoder'],
        config['_semantic_map_classes'], data['rasters']['semantics'],
    )
    f = pyr['features'][-1]
    plane = dict(features=f, valid=np.ones(f.shape[:-1], bool))
    pred['semantic'] = {'feature_plane': plane}
    planes.append(plane)
  pred['bev_features'] = plane = fuse_neural_maps(config, planes, params.get('modality_fusion'))
  if config.get('matching_dim') is not None:
    pred['bev_matching'] = matching_head(params, config, plane)
  if config.get('add_confidence'):
    # bev_mapper.py:292-295
    head = params['confidence_head']['layers_0']
    scores = (plane['features'] @ head['kernel'])[..., 0] + head['bias'][0]
    conf = log_sigmoid(scores.astype(np.float32))
    pred['bev_confidence'] = np.where(plane['valid'], conf, 0).astype(np.float32)
  pred['_xyz_query'] = data.get('xyz_query')
  return pred


# snap/data/types.py:35-42
SURFEL_ROAD_CLASSES = ('crosswalk', 'sidewalk', 'pavedroad', 'stopline', 'line', 'otherlanemarking')


def semantic_raster_embed(params, raster_classes, rasters, dtype=np.float32):
  """semantic_raster_encoder.py:33-46,63-79: rasters [..., N] bool -> [..., (1 + n_other) * E]."""
  idx_road = [i for i, c in enumerate(raster_classes) if c in SURFEL_ROAD_CLASSES]
  idx_other = [i for i, c in enumerate(raster_classes) if c not in SURFEL_ROAD_CLASSES]
  t_road = params['embeddings_surfel_road']['embedding']
  t_other = params['embeddings_other_classes']['embedding']
  label_road = np.argmax(rasters[..., idx_road], axis=-1)                 # :65-66
  f_road = t_road[label_road]                                               # :67
  labels_other = np.arange(len(idx_other)) + rasters[..., idx_other].astype(int)   # :70-72 (sic)
  f_other = t_other[labels_other]                                           # :73
  f_other = f_other.reshape(*f_other.shape[:-2], -1)                        # :75
  return np.concatenate([f_road, f_other], axis=-1).astype(dtype)          # :77


def semantic_raster_encoder(params, config, raster_classes, rasters):
  f = semantic_raster_embed(params, raster_classes, rasters,
                            params['embeddings_surfel_road']['embedding'].dtype)
  return encoder.image_encoder(params['encoder'], config['encoder'], f)      # :78
