# two-stream kernel trace of the default bench (prefetch on): per-stream busy time, union, gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2_tl_bench.json 2>/dev/null
cp /tmp/tl/*/*kernel_trace.csv gpurun_out/r2_tl_kernel_trace.csv
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/tl/*/*kernel_trace.csv')[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Stream_Id'] if 'Stream_Id' in r else r['Queue_Id'],r['Kernel_Name'][:60]) for r in csv.DictReader(open(f))]
rows.sort()
t0,t1=rows[0][0],max(r[1] for r in rows)
# steady state: the last 100 ms of kernel activity (~ 6-7 steps of the timed region)
cut=t1-100e6
rows=[r for r in rows if r[0]>=cut]
t0=rows[0][0]; t1=max(r[1] for r in rows)
per=collections.defaultdict(float)
for s,e,q,n in rows: per[q]+=e-s
def union(iv):
    iv=sorted(iv); tot=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: tot+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    return tot+ce-cs
u=union([(s,e) for s,e,_,_ in rows])
print("window %.2f ms, kernels %d" % ((t1-t0)/1e6, len(rows)))
for q,v in sorted(per.items(), key=lambda kv:-kv[1]): print("  stream/queue %s: sum of kernel time %.2f ms, union %.2f ms" % (q, v/1e6, union([(s,e) for s,e,qq,_ in rows if qq==q])/1e6))
print("union over all streams %.2f ms -> idle %.2f ms (%.1f %%)" % (u/1e6, (t1-t0-u)/1e6, 100*(t1-t0-u)/(t1-t0)))
# gap histogram on the busiest stream
both=union([(s,e) for s,e,_,_ in rows])
ov=sum(per.values())-both
print("sum of both streams %.2f ms, concurrent (overlapped) time %.2f ms" % (sum(per.values())/1e6, ov/1e6))
qb=max(per,key=per.get)
iv=sorted((s,e) for s,e,qq,_ in rows if qq==qb)
gaps=[iv[i+1][0]-iv[i][1] for i in range(len(iv)-1)]
import statistics
g=[x for x in gaps if x>0]
print("busiest stream: %d gaps, total %.2f ms, median %.1f us, >20us: %d totalling %.2f ms" % (len(g), sum(g)/1e6, statistics.median(g)/1e3, sum(1 for x in g if x>20000), sum(x for x in g if x>20000)/1e6))
PY
