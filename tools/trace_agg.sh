# kernel trace of the default bench under the given env assignments, aggregated per (stream, kernel, grid) over the last
# 100 ms: usage trace_agg.sh <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
tag=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/ta_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/ta_$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $BENCH_ARGS > gpurun_out/r2_ta_${tag}_bench.json 2>/dev/null
python3 - $tag <<'PY'
import csv,glob,collections,re,sys
tag=sys.argv[1]
f=glob.glob('/tmp/ta_%s/*/*kernel_trace.csv'%tag)[0]
rows=list(csv.DictReader(open(f)))
t1=max(int(r['End_Timestamp']) for r in rows)
rows=[r for r in rows if int(r['Start_Timestamp'])>=t1-100e6]
def nm(r):
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'^void ','',n); return re.sub(r'[(].*','',n)[:90]
agg=collections.defaultdict(lambda:[0,0])
for r in rows:
    k=(r['Stream_Id'],nm(r),r['Grid_Size_X'],r['Workgroup_Size_X'])
    agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
w=csv.writer(open('gpurun_out/r2_ta_%s_agg.csv'%tag,'w'))
w.writerow(['stream','kernel','grid','wg','count','total_ns'])
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): w.writerow(list(k)+[c,t])
per=collections.Counter()
for k,(c,t) in agg.items(): per[k[0]]+=t
print(tag, {s:round(v/1e6,2) for s,v in per.items()}, "ms of kernel time per stream in the last 100 ms")
PY
