# HIP-API trace + kernel trace of the default bench: for every kernel, how long before its GPU start was the launch API
# call made (queue lead)?  A lead near zero means the GPU was waiting for the host at that point.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/tl2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2_tl2_bench.json 2>/dev/null
ls /tmp/tl2/*/ > gpurun_out/r2_tl2_files.txt
python3 - <<'PY'
import csv,glob,collections,re
kf=glob.glob('/tmp/tl2/*/*kernel_trace.csv')[0]
af=glob.glob('/tmp/tl2/*/*hip_api_trace.csv')[0]
K=list(csv.DictReader(open(kf))); A=list(csv.DictReader(open(af)))
print("kernel cols", list(K[0].keys())); print("api cols", list(A[0].keys()))
api={r['Correlation_Id']:r for r in A}
t1=max(int(r['End_Timestamp']) for r in K)
K=[r for r in K if int(r['Start_Timestamp'])>=t1-100e6]
K.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r):
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'^void ','',n); return r['Stream_Id']+':'+re.sub(r'[<(].*','',n)[:34]
lead=collections.defaultdict(list)
prev_end={}
out=open('gpurun_out/r2_tl2_leads.csv','w'); out.write("stream,kernel,api,api_start,api_end,k_start,k_end,gap_before_us,lead_us\n")
for r in K:
    a=api.get(r['Correlation_Id'])
    if a is None: continue
    s=r['Stream_Id']; ks=int(r['Start_Timestamp'])
    gap=(ks-prev_end[s])/1e3 if s in prev_end else 0.0
    prev_end[s]=max(prev_end.get(s,0),int(r['End_Timestamp']))
    ld=(ks-int(a['End_Timestamp']))/1e3
    lead[nm(r)].append((gap,ld))
    out.write("%s,%s,%s,%s,%s,%s,%s,%.1f,%.1f\n"%(s,nm(r),a['Function'],a['Start_Timestamp'],a['End_Timestamp'],r['Start_Timestamp'],r['End_Timestamp'],gap,ld))
out.close()
print("%-40s %6s %12s %12s %14s"%("kernel","n","med lead us","min lead us","lead@gaps>30us"))
for k,v in sorted(lead.items(), key=lambda kv:-len(kv[1]))[:40]:
    ls=sorted(l for g,l in v); big=[l for g,l in v if g>30]
    print("%-40s %6d %12.0f %12.0f %14s"%(k,len(v),ls[len(ls)//2],ls[0],("%.0f (n=%d)"%(sorted(big)[len(big)//2],len(big))) if big else "-"))
PY
