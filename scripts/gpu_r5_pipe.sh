#!/bin/bash
# A/B of the scene pipeline (chunk sizes) on the C2 bench + an equality check of its outputs.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r05p
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=.
python - <<'PY' 2>&1 | tail -12
import torch, bench
from snap_amd import ops
dev = torch.device('cuda', 0)
loc, cfg, meta, variables, batch = bench.build('c2', dev, 0, materialize_volume=False)
loc.engine = 'bf16x3'
out = {}
for n in (0, 2, 1, 4):
  ops.SCENE_PIPELINE_CHUNK = n
  p = loc.apply(variables, batch, train=False, rngs={'sampling': 7})
  torch.cuda.synchronize()
  out[n] = p
ref = out[0]
for n in (2, 1, 4):
  p = out[n]
  print('chunk', n,
        'bev_matching map equal', torch.equal(p['map']['bev_matching'].features, ref['map']['bev_matching'].features),
        'query equal', torch.equal(p['query']['bev_matching'].features, ref['query']['bev_matching'].features),
        'scores max|d|', float((p['scores_poses'] - ref['scores_poses']).abs().max()),
        'best equal', torch.equal(p['best_index'], ref['best_index']),
        'pyr shapes', [tuple(f.shape) for f in p['map']['streetview']['image_feature_pyramid'].features] ==
                      [tuple(f.shape) for f in ref['map']['streetview']['image_feature_pyramid'].features],
        'pyr equal', all(torch.equal(a, b) for a, b in zip(p['map']['streetview']['image_feature_pyramid'].features,
                                                           ref['map']['streetview']['image_feature_pyramid'].features)))
PY
for n in 0 2 1 4; do
  SNAP_SCENE_PIPELINE_CHUNK=$n timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 > $O/bench_chunk$n.json
  python -c "
import json; d=json.load(open('$O/bench_chunk$n.json')); print('chunk $n', d['ms_per_step'], d['step_ms'])"
done
