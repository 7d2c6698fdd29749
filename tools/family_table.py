"""Per-layer table of the GEMM families of a bench dump (SNAP_BENCH_DUMP): launches of one shape aggregated,
sorted by time.  python tools/family_table.py <launches.json>"""
import json,sys
d=json.load(open(sys.argv[1]))
for fam in ('conv_wgrad_bf16','conv_wgrad_fp16','conv_bf16','conv_fp16','conv_wgrad'):
    L=d.get(fam,[])
    agg={}
    for name,ms,fl,by in L:
        a=agg.setdefault(name,[0,0.0,fl])
        a[0]+=1; a[1]+=ms
    print(fam, len(L), round(sum(x[1] for x in L),3))
    for name,(n,ms,fl) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:60]:
        print('  %-48s x%-3d %8.3f ms  avg %7.1f us  %6.1f TF/s' % (name,n,ms,ms/n*1e3, (fl or 0)/(ms/n)/1e9 if ms else 0))
