cd /root/repo; export PYTHONPATH=. TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py -m gpu -q -x -k "pose_score or c4_pose or lattice or planted" 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C4', d['ms_per_step'], {n: round(v['ms'],3) for n,v in d['kernels'].items() if v['ms']>0.05})"; done
