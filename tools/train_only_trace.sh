# clean kernel profile of the training step alone (tools/train_only_bench.py: prebuilt batches, one stream)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/to_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/to_tr -- python3 tools/train_only_bench.py 20 > gpurun_out/r2_to_trace.log 2>/dev/null
python3 - <<'PY'
import csv,glob,collections,re
f=glob.glob('/tmp/to_tr/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last 20 steps: find by counting sgd_step_kernel launches from the end
ends=[i for i,r in enumerate(rows) if 'sgd_step_kernel' in r['Kernel_Name']]
first=ends[-21]+1; last=ends[-1]
rows=rows[first:last+1]
steps=20.0
t0=int(rows[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in rows)
def nm(r):
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'^void ','',n); return re.sub(r'[(].*','',n)[:70]
agg=collections.defaultdict(lambda:[0,0])
busy=0
for r in rows:
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp']); busy+=d
    k=nm(r); agg[k][0]+=1; agg[k][1]+=d
print("wall %.3f ms/step, kernel time %.3f ms/step, gaps %.3f ms/step, launches %.1f/step"%((t1-t0)/steps/1e6,busy/steps/1e6,(t1-t0-busy)/steps/1e6,len(rows)/steps))
gaps=[int(b['Start_Timestamp'])-int(a['End_Timestamp']) for a,b in zip(rows,rows[1:])]
import statistics
print("gap median %.1f us, mean %.1f us, >20us: %d/step totalling %.3f ms/step"%(statistics.median(gaps)/1e3,statistics.mean(gaps)/1e3,sum(1 for g in gaps if g>20000)/steps,sum(g for g in gaps if g>20000)/steps/1e6))
w=csv.writer(open('gpurun_out/r2_to_agg.csv','w')); w.writerow(['kernel','n_per_step','ms_per_step','avg_us'])
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    w.writerow([k,c/steps,t/steps/1e6,t/c/1e3])
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]:
    print("  %-72s n/step %5.1f %7.3f ms/step avg %7.1f us"%(k,c/steps,t/steps/1e6,t/c/1e3))
PY
