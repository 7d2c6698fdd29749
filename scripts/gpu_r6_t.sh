cd /root/repo; export PYTHONPATH=. TMPDIR=/tmp
for lib in "" alt_lift4 alt_lift6 ""; do
SNAP_HIP_LIB=${lib:+$PWD/snap_amd/lib/$lib/libsnap_hip.so} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2 ${lib:-default}', d['ms_per_step'], d['step_ms']['median'], {n: round(v['ms'],3) for n,v in d['kernels'].items() if n in ('lift_pool','mlp2_pool_bf16x3')})"; done
