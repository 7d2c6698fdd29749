"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected SEPARATELY over
`bench.py --steps 1 --warmup 1`) into profiles/<name>_hbm_traffic.json, the file
bench.py's `roofline.traffic` reads.

  python tools/make_hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [steps]

Units: both counters report KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE reports half of wide coalesced reads -> read bytes = 2 x FETCH_SIZE x 1024
(exact for the 16-B/lane streaming kernels, an upper bound for strided access);
write bytes = WRITE_SIZE x 1024.
"""
import collections
import csv
import json
import re
import sys

# kernel-name substring -> bench.py kernel family (the KernelProfiler region names)
FAMILIES = [
    ('conv_igemm_kernel', 'conv_igemm'), ('conv_split_kernel', 'conv_split'), ('conv3x3_halo_kernel', 'conv_split'),
    ('splitk_reduce_kernel', 'conv_split'), ('conv_split_root_kernel', 'conv_split'),
    ('conv1x1_rs_kernel', 'conv_split'), ('conv1x1_bs_kernel', 'conv_split'), ('conv3x3_ws64_kernel', 'conv_split'), ('conv_root_ws64_kernel', 'conv_split'),
    ('splitk_reduce_stats_kernel', 'conv_split'), ('conv_ps_kernel', 'conv_split'),
    ('gn_norm_split_kernel', 'gn_norm_split'), ('presplit_kernel', 'presplit'),
    ('gn_finalize_tiled_kernel', 'group_norm_stats'), ('sim_split_kernel', 'sim_softmax'),
    ('sim_split_fast_kernel', 'sim_softmax'), ('sim_presplit_kernel', 'sim_softmax'),
    ('ransac_sample', 'ransac_sample'), ('template_', 'voting'), ('conv_bf16_kernel', 'conv_bf16'),
    ('mlp2_pool_kernel', 'mlp2_pool'), ('mlp2_pool_finalize_kernel', 'mlp2_pool'), ('fill_f32_kernel', 'mlp2_pool'),
    ('pack_weights_split', 'pack_weights'),
    ('pose_score_db_kernel', 'pose_score'), ('pose_score_kernel', 'pose_score'),
    ('pose_table', 'pose_score'), ('pose_score_reduce', 'pose_score'),
    ('lift_pool_batched_kernel', 'lift_pool'), ('lift_pool_kernel', 'lift_pool'),
    ('vertical_pool_kernel', 'vertical_pool'), ('vertical_pool_wave_kernel', 'vertical_pool'),
    ('gn_partial_kernel', 'group_norm_stats'), ('gn_finalize_kernel', 'group_norm_stats'),
    ('sim_kernel', 'sim_softmax'), ('sim_mfma_kernel', 'sim_softmax'), ('row_stats_kernel', 'sim_softmax'),
    ('ransac_sample_kernel', 'ransac_sample'), ('chunk_prefix_kernel', 'ransac_sample'), ('weight_std', 'weight_standardize'),
    ('plane_fuse_match', 'plane_fuse_match'), ('max_pool_kernel', 'max_pool'),
    ('count_rows_kernel', 'compact_rows'), ('scan_blocks_kernel', 'compact_rows'),
    ('write_rows_kernel', 'compact_rows'), ('fill_masked_rows_kernel', 'fill_masked_rows'),
]


def short(name):
  name = name.replace('(anonymous namespace)::', '').replace('void ', '')
  return re.split(r'\(', name)[0][:90]


def read(path, counter):
  tot = collections.defaultdict(float)
  cnt = collections.Counter()
  for r in csv.DictReader(open(path)):
    if r['Counter_Name'] != counter:
      continue
    k = short(r['Kernel_Name'])
    tot[k] += float(r['Counter_Value'])
    cnt[k] += 1
  return tot, cnt


def main():
  fetch_csv, write_csv, out = sys.argv[1:4]
  steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
  f, fc = read(fetch_csv, 'FETCH_SIZE')
  w, _ = read(write_csv, 'WRITE_SIZE')
  kernels = {}
  per_step = collections.defaultdict(lambda: {'hbm_read_bytes': 0.0, 'hbm_write_bytes': 0.0, 'launches': 0, 'body_launches': 0})
  for k in sorted(set(f) | set(w)):
    kernels[k] = {f'launches_{steps}steps': fc.get(k, 0), 'FETCH_SIZE_KiB': f.get(k, 0.0),
                  'WRITE_SIZE_KiB': w.get(k, 0.0)}
    fam = next((fam for sub, fam in FAMILIES if sub in k), None)
    if fam is None:
      fam = 'other'
    per_step[fam]['hbm_read_bytes'] += 2.0 * f.get(k, 0.0) * 1024 / steps
    per_step[fam]['hbm_write_bytes'] += w.get(k, 0.0) * 1024 / steps
    per_step[fam]['launches'] += fc.get(k, 0) // steps
    # (the launches bench.py's event regions count: not the reduce / finalize / fill passes behind them)
    if not any(t in k for t in ('reduce', 'finalize', 'fill_', 'presplit')):
      per_step[fam]['body_launches'] += fc.get(k, 0) // steps
  note = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes over `bench.py --steps 1 '
          f'--warmup 1` ({steps} steps per pass), C2 workload. Units KiB. read bytes = 2 x FETCH_SIZE '
          'x 1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads, MI355X_MICROARCH.md HBM '
          'section; exact for 16-B/lane streaming kernels, an upper bound for strided access); write '
          'bytes = WRITE_SIZE x 1024. per_step = bytes of ONE bench step per kernel family '
          '(tools/make_hbm_traffic.py).')
  # what bench.py's `roofline.traffic_stale` compares against: the sources the passes ran on (the
  # digest is computed on the box; the git head is handed in through SNAP_GIT_HEAD -- .git does not
  # travel) and the launches of a family in one step
  import os
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import bench
  json.dump({'_note': note, 'git_head': os.environ.get('SNAP_GIT_HEAD'), 'source_digest': bench.source_digest(),
             'kernels': kernels, 'per_step': per_step}, open(out, 'w'), indent=1)
  for fam in ('conv_igemm', 'conv_split', 'pose_score', 'lift_pool', 'sim_softmax'):
    if fam in per_step:
      print(fam, {k: f'{v / 1e9:.3f} GB' for k, v in per_step[fam].items()})


if __name__ == '__main__':
  main()
