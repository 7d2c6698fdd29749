"""Generates the golden fixtures in this directory from the numpy oracle.

  python tests/golden/make_golden.py

Fixtures are DATA (seeded inputs + oracle outputs as .npz): regression pins for the
oracle itself (``-m "not gpu"``) and fixed vectors for the HIP kernels (``-m gpu``).
The reference ships no golden vectors and cannot run offline; these
come from the oracle, which is pinned against scipy / torch-CPU / analytic answers in
tests/test_oracle_pins.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_ops  # noqa: E402
from oracle import grids as o_grids  # noqa: E402
from oracle import voting as o_voting  # noqa: E402
from snap_amd.data import synthetic  # noqa: E402
from snap_amd.utils import grids  # noqa: E402


def t(a):
  return torch.as_tensor(np.ascontiguousarray(a))


def save(name, **arrays):
  out = {}
  for k, v in arrays.items():
    out[k] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
  path = os.path.join(HERE, name + '.npz')
  np.savez_compressed(path, **out)
  print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def main():
  rng = np.random.default_rng(2024)
  f32 = lambda *s: rng.standard_normal(s).astype(np.float32)

  # (ix) encoder block: GN statistics + fused conv + FPN up-sample-add on 8x8 inputs.
  x = t(f32(2, 8, 8, 64) * 1.5 + 0.3)
  gamma, beta = t(f32(64) * 0.3 + 1), t(f32(64) * 0.1)
  w3 = t(f32(3, 3, 64, 32) / 24)
  mu, sc = oracle_ops.group_norm_stats(x, gamma)
  y = oracle_ops.conv2d(x, w3, padding=((1, 1), (1, 1)), prologue=2, gn=(mu, sc, beta))
  mu2, sc2 = oracle_ops.group_norm_stats(x, gamma, relu_first=True)
  prev = t(f32(2, 4, 4, 32))
  w1 = t(f32(1, 1, 64, 32) / 8)
  yf = oracle_ops.conv2d(x, w1, prologue=3, gn=(mu2, sc2, beta), up_prev=prev)
  w_in = t(f32(3, 3, 16, 8) * 0.2 + 0.1)
  ws = oracle_ops.weight_standardize(w_in)
  save('encoder_block', x=x, gamma=gamma, beta=beta, w3=w3, mu=mu, sc=sc, y=y, mu_relu=mu2,
       sc_relu=sc2, prev=prev, w1=w1, y_fpn=yf, w_std_in=w_in,
       w_std=ws)

  # (ii)-(iv) lift: 4x4x3 grid, V=3, K=2, fisheye k != 0, points behind cameras included.
  g = grids.Grid3D((4, 4, 3), 0.8)
  batch = synthetic.make_batch(1, g, 3, (24, 32), seed=5, with_aerial=False, with_gt=False,
                               k_radial=0.03)
  cam = batch['map']['camera'].scale(torch.tensor([0.25, 0.25])).packed()
  Rt = batch['map']['T_view2scene'].packed()
  pts = t(np.stack([rng.uniform(-1, 4, (1, 48)), rng.uniform(-1, 4, (1, 48)),
                    rng.uniform(-1, 3, (1, 48))], -1).astype(np.float32))
  f = t(f32(1, 3, 6, 8, 12))
  kw = dict(fisheye=True, feature_dim=8, num_bins=4, depth_min_max=(0.5, 8.0))
  p2d, vis, depth = oracle_ops.project_points(cam, Rt, pts, True)
  pooled_k2, valid_k2 = oracle_ops.lift_pool(f, cam, Rt, pts, K=2, **kw)
  pooled_all, valid_all = oracle_ops.lift_pool(f, cam, Rt, pts, K=0, **kw)
  save('lift', f=f, cam=cam, Rt=Rt, pts=pts, p2d=p2d, vis=vis, depth=depth, pooled_k2=pooled_k2,
       valid_k2=valid_k2, pooled_all=pooled_all, valid_all=valid_all)

  # (v)-(vi) BEV: vertical pooling with fully-invalid columns, fuse + matching incl. zero norm.
  vol = t(f32(1, 5, 4, 6, 16))
  vv = t(rng.random((1, 5, 4, 6)) > 0.5)
  vv[0, 0] = False
  plane, pvalid = oracle_ops.vertical_pool(vol, vv)
  aerial = t(f32(1, 5, 4, 16))
  Wm, bm = t(f32(16, 8) * 0.2), t(np.zeros(8, np.float32))
  plane0 = plane.clone()
  plane0[0, 1, 1] = 0
  aerial[0, 1, 1] = 0
  fused, fvalid, matching = oracle_ops.plane_fuse_match([plane0, aerial], [pvalid, None], 'max', Wm, bm)
  save('bev', vol=vol, vvalid=vv, plane=plane, pvalid=pvalid, plane0=plane0, aerial=aerial, Wm=Wm,
       bm=bm, fused=fused, fvalid=fvalid, matching=matching)

  # (vii) pose: similarity, scoring (oob on/off), refinement lattice.
  B, Nq, X, Y, Dm, P = 1, 24, 12, 10, 8, 40
  fq = f32(B, Nq, Dm); fq /= np.linalg.norm(fq, axis=-1, keepdims=True)
  fm = f32(B, X, Y, Dm); fm /= np.linalg.norm(fm, axis=-1, keepdims=True)
  nv = t(np.array([Nq - 2], np.float32))
  sim, stats, prob, rowstats = oracle_ops.sim_softmax(t(fq), t(fm), float(np.exp(2.0)), True, nv,
                                                      want_prob=True)
  poses = t(np.stack([rng.uniform(-3, 3, (B, P)), rng.uniform(-1, 4, (B, P)),
                      rng.uniform(-1, 3, (B, P))], -1).astype(np.float32))
  q_xy = t(rng.uniform(-1, 1, (B, Nq, 2)).astype(np.float32))
  vq = t(rng.random((B, Nq)) > 0.1)
  mv = t(rng.random((B, X, Y)) > 0.1)
  s0 = oracle_ops.pose_score(sim, poses, q_xy, vq, mv, 0.25, mask_oob=False)
  s1 = oracle_ops.pose_score(sim, poses, q_xy, vq, mv, 0.25, mask_oob=True)
  corr = t(np.stack([rng.integers(0, Nq, (B, 2 * 3 * 5)), rng.integers(0, X, (B, 30)),
                     rng.integers(0, Y, (B, 30))], -1).astype(np.int32))
  kposes = oracle_ops.poses_from_corr(corr, q_xy, 5, 3, 0.25)
  save('pose', fq=fq, fm=fm, num_valid=nv, sim=sim, stats=stats, prob=prob, poses=poses, q_xy=q_xy,
       valid_q=vq, map_valid=mv, scores=s0, scores_oob=s1, corr=corr, kabsch_poses=kposes)

  # (viii) exhaustive voting at 16^2, R=8, partial validity.
  H, R, D = 16, 8, 4
  og = o_grids.Grid2D((H, H), 0.5)
  q = f32(H, H, D); qv = rng.random((H, H)) > 0.15; q *= qv[..., None]
  m = f32(H, H, D); mvv = rng.random((H, H)) > 0.1
  templates, tvalid = o_voting.sample_query_templates(q, qv, R, og)
  scores = o_voting.template_matching(templates, tvalid, m, mvv)
  save('voting', q=q, q_valid=qv, m=m, m_valid=mvv, templates=templates, tvalid=tvalid, scores=scores)

  # (xi) later additions: bf16-operand conv (rounded-operand restatement), ViT pieces, semantic embed.
  xb = t(f32(2, 6, 5, 40) + 0.2)
  wb = t(f32(3, 3, 40, 24) / 19)
  gam, bet = t(f32(40) * 0.3 + 1), t(f32(40) * 0.1)
  mub, scb = oracle_ops.group_norm_stats(xb, gam, groups=8)
  yb = oracle_ops.conv2d(xb, wb, padding=((1, 1), (1, 1)), prologue=2, gn=(mub, scb, bet), math='bf16')
  xd = t(f32(50, 36)); wd = t(f32(33, 16) / 6); bd = t(f32(16))
  yd = oracle_ops.dense(xd, wd, bd, cin=33, gelu=True, math='bf16')
  save('bf16', x=xb, w=wb, beta=bet, mu=mub, sc=scb, y=yb, xd=xd, wd=wd, bd=bd, yd=yd)

  xl = t(f32(11, 192) * 1.4 + 0.3)
  gl, bl = t(f32(192) * 0.3 + 1), t(f32(192) * 0.2)
  yl = oracle_ops.layer_norm(xl, gl, bl)
  qkv = t(f32(2, 70, 3, 2, 64))
  att = oracle_ops.attention(qkv)
  save('vit', x=xl, gamma=gl, beta=bl, y=yl, qkv=qkv, att=att)

  ras = t(rng.random((2, 6, 5, 7)) < 0.4)
  t_road, t_other = t(f32(3, 8)), t(f32(8, 8))
  emb = oracle_ops.semantic_embed(ras, [0, 2, 5], [1, 3, 4, 6], t_road, t_other)
  save('semantic', rasters=ras, table_road=t_road, table_other=t_other, out=emb)


if __name__ == '__main__':
  main()
