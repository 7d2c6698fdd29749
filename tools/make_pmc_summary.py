"""Per-kernel-family utilisation summary from rocprofv3 PMC passes over one bench step.

  python tools/make_pmc_summary.py <out.json> <pass1_counter_collection.csv> [<pass2...> ...]

Counters expected (any subset): SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES,
SQ_INSTS_VALU, SQ_WAIT_INST_ANY, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, GRBM_GUI_ACTIVE.
Normalisations (MI355X: 8 XCD x 4 SE, 8 CU / SE, 4 SIMD / CU):
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32 SIMDs per SE)
  valu_busy   = SQ_INSTS_VALU x 4 cycles / (SQ_BUSY_CYCLES x 32)
  lds_busy    = SQ_LDS_IDX_ACTIVE / (SQ_BUSY_CYCLES x 8 CUs per SE)
  lds_conflict= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_frac   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
"""
import collections
import csv
import json
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from make_hbm_traffic import FAMILIES, short  # noqa: E402


def main():
  out = sys.argv[1]
  agg = collections.defaultdict(lambda: collections.defaultdict(float))
  for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
      k = short(r['Kernel_Name'])
      fam = next((fam for sub, fam in FAMILIES if sub in k), None)
      if fam is None:
        continue
      agg[fam][r['Counter_Name']] += float(r['Counter_Value'])
  res = {}
  for fam, v in agg.items():
    busy = v.get('SQ_BUSY_CYCLES', 0.0)
    e = {}
    if busy:
      if 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        e['mfma_busy'] = round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (busy * 32), 3)
      if 'SQ_INSTS_VALU' in v:
        e['valu_busy'] = round(v['SQ_INSTS_VALU'] * 4 / (busy * 32), 3)
      if 'SQ_LDS_IDX_ACTIVE' in v:
        e['lds_busy'] = round(v['SQ_LDS_IDX_ACTIVE'] / (busy * 8), 3)
    if v.get('SQ_LDS_IDX_ACTIVE'):
      e['lds_conflict'] = round(v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE'], 3)
    if v.get('SQ_WAVE_CYCLES') and 'SQ_WAIT_INST_ANY' in v:
      e['wait_frac'] = round(v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'], 3)
    res[fam] = e
  note = ('rocprofv3 --pmc passes (counters only) over `bench.py --steps 1 --warmup 1`, C2; per kernel '
          'family; normalisations in tools/make_pmc_summary.py')
  json.dump({'_note': note, 'families': res}, open(out, 'w'), indent=1)
  for fam, e in sorted(res.items()):
    print(fam, e)


if __name__ == '__main__':
  main()
