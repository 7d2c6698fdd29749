cd /root/repo
export PYTHONPATH=.
for v in 0 1 0 1; do
  SNAP_OVERLAP_QUERY=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('overlap_query=$v', d['ms_per_step'], d['step_ms'])"
done
