cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "lift" --timeout=600 2>&1 | tail -3
for lib in "" "snap_amd/lib/alt/libsnap_hip.so" "" "snap_amd/lib/alt/libsnap_hip.so"; do echo "== lib=${lib:-default}";
 SNAP_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(d['ms_per_step'], {n:round(v['ms'],2) for n,v in k.items() if v['ms'] > 0.3})"; done
