# kernel stats of one workload of bench.py: usage trace_workload.sh <workload> [steps]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
W=${1:-dales_deform}; S=${2:-4}
rm -rf /tmp/tw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tw -- python3 bench.py --workload $W --steps $S --warmup 2 --no-cpu-baseline --distinct-batches 2 > gpurun_out/r2_tw_${W}.json 2>/dev/null
cp /tmp/tw/*/*kernel_stats.csv gpurun_out/r2_tw_${W}_stats.csv
python3 - $W <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open('gpurun_out/r2_tw_%s_stats.csv'%sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel time %.1f ms over %d launches"%(tot/1e6,sum(int(r['Calls']) for r in rows)))
for r in rows[:28]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n); n=re.sub(r'\(.*','',n)[:84]
    print("  %-86s calls %5s  %8.2f ms  %5.1f %%  avg %8.1f us"%(n,r['Calls'],int(r['TotalDurationNs'])/1e6,100*int(r['TotalDurationNs'])/tot,float(r['AverageNs'])/1e3))
PY
