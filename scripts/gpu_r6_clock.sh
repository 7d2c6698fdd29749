#!/bin/bash
# round 6: effective shader clock per kernel (GRBM_GUI_ACTIVE / duration) inside a C2 step
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
R=$PWD; O=gpurun_out/r6clk; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/p" -o pmc -- \
  python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --in-flight 1) > $O/log.txt 2>&1
ls $O/p
python - <<PY
import csv, collections, glob
tr={}
for f in glob.glob('$O/p/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        tr[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']), r['Kernel_Name'])
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for f in glob.glob('$O/p/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']!='GRBM_GUI_ACTIVE': continue
        d=tr.get(r['Dispatch_Id'])
        if not d: continue
        a=agg[d[1][:70]]; a[0]+=1; a[1]+=d[0]; a[2]+=float(r['Counter_Value'])
rows=sorted(agg.items(), key=lambda kv:-kv[1][1])[:30]
for k,(n,ns,cyc) in rows:
    print(f"{k:70s} n={n:4d} {ns/1e6:8.3f} ms  clk {cyc/ns:6.3f} GHz-equiv (cycles/ns)")
tot_ns=sum(v[1] for v in agg.values()); tot_c=sum(v[2] for v in agg.values())
print('overall', tot_c/tot_ns)
PY
rm -rf $O/p
