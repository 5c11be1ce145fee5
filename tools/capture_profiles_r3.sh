# tools/capture_profiles_r3.sh <name> -- run on the GPU box (gpurun): round-3 evidence.  Bench lines of every workload / mode,
# rocprofv3 kernel stats (two streams and one), FETCH_SIZE / WRITE_SIZE per kernel in SEPARATE --pmc passes (never combined with
# other trace domains), for the training step (config 3), the forward-only pass and config 5.  Output: gpurun_out/<name>/;
# the files judged are copied to profiles/ by hand afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-r03}
mkdir -p $O
b() { python3 bench.py "$@" 2> $O/last.err; }
b > $O/bench_dales_1gpu.json; echo bench dales done
b --contrast 0 --no-cpu-baseline > $O/bench_dales_no_contrast_1gpu.json
b --workload vaihingen --no-cpu-baseline > $O/bench_vaihingen_1gpu.json
b --workload vaihingen_wl --no-cpu-baseline > $O/bench_vaihingen_wl_1gpu.json
b --workload dales_deform --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_dales_deform_bf16_1gpu.json
b --mode infer --steps 30 --no-cpu-baseline > $O/bench_dales_infer_1gpu.json
b --mode infer --steps 30 --no-cpu-baseline --prefetch 0 > $O/bench_dales_infer_noprefetch_1gpu.json
echo benches done
stats() {   # <tag> <bench args...>
  tag=$1; shift
  rm -rf /tmp/ks_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python3 bench.py "$@" --no-cpu-baseline > $O/ks_$tag.json 2> /dev/null
  cp /tmp/ks_$tag/*/*kernel_stats.csv $O/kernel_stats_$tag.csv
  echo stats $tag done
}
stats dales_5steps --steps 5 --warmup 2
stats dales_5steps_noprefetch --steps 5 --warmup 2 --prefetch 0
stats dales_deform_4steps --workload dales_deform --steps 4 --warmup 2
stats dales_deform_4steps_noprefetch --workload dales_deform --steps 4 --warmup 2 --prefetch 0
stats vaihingen_20steps --workload vaihingen --steps 20 --warmup 5
stats dales_infer_10steps_noprefetch --mode infer --steps 10 --warmup 2 --prefetch 0
pmc() {     # <tag> <bench args...>
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_${tag}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_${tag}_$c -- python3 bench.py "$@" --no-cpu-baseline --prefetch 0 > /dev/null 2>&1
    python3 - /tmp/pm_${tag}_$c $c >> $O/pmc_per_kernel_$tag.csv <<'PY'
import csv,glob,sys,re,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name'])
        n=re.sub(r'\(.*','',n)[:110]
        d[n].append(float(r['Counter_Value']))
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:45]:
    print('%s,"%s",%d,%.1f,%.1f' % (sys.argv[2], n, len(v), sum(v)/len(v), max(v)))
PY
    echo pmc $tag $c done
  done
}
pmc dales_train --steps 2 --warmup 1 --contrast 0
pmc dales_infer --mode infer --steps 2 --warmup 1
pmc dales_deform --workload dales_deform --steps 2 --warmup 1
ls -la $O
