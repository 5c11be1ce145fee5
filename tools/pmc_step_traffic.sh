# tools/pmc_step_traffic.sh -- on the GPU box: HBM traffic of one DALES training step (one stream, contrastive term included) from
# separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (kernel trace only), with the epilogue folding on (default)
# and off (the four A/B switches).  Output: gpurun_out/pmc_step/{on,off}_per_kernel.csv (counter, kernel, launches, avg, max:
# raw counter values, KB) and a summary line per build: raw KB per step (sum over all kernels / number of sgd_step launches).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/pmc_step; mkdir -p $O; rm -f $O/*.csv $O/summary.txt
run() {   # <tag> <env assignments...>
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ps_${tag}_$c
    env "$@" rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ps_${tag}_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --prefetch 0 > /dev/null 2>&1
    python3 - /tmp/ps_${tag}_$c $c $tag >> $O/summary.txt 3>> $O/${tag}_per_kernel.csv <<'PY'
import csv,glob,sys,re,collections,os
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name'])
        n=re.sub(r'\(.*','',n)[:110]
        d[n].append(float(r['Counter_Value']))
steps=max(1,len([1 for n,v in d.items() if 'sgd_step' in n for _ in v]))
tot=sum(sum(v) for v in d.values())
print('%s %s: %.1f MB raw per step over %d steps, %d launches per step' % (sys.argv[3], sys.argv[2], tot/steps/1024.0, steps, sum(len(v) for v in d.values())//steps))
out=os.fdopen(3,'w')
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:50]:
    out.write('%s,"%s",%d,%.1f,%.1f\n' % (sys.argv[2], n, len(v), sum(v)/len(v), max(v)))
PY
  done
}
run on X=1
run off WEASAL_DROPOUT_FUSED=0 WEASAL_SKIP_SLOTS=0 WEASAL_BLOCK_GATHER_RESIDUAL=0 WEASAL_GATE_LINKS=0
cat $O/summary.txt
