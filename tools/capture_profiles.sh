# tools/capture_profiles.sh <name> -- run on the GPU box (gpurun): bench JSON of the three workloads, rocprofv3 kernel stats
# (two streams and one), FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes) into gpurun_out/<name>/; the files judged
# are copied to profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-capture}
mkdir -p $O
python3 bench.py > $O/bench_dales.json 2> $O/bench_dales.err
echo bench dales done
python3 bench.py --workload vaihingen --no-cpu-baseline > $O/bench_vaihingen.json 2> $O/bench_vaihingen.err
echo bench vaihingen done
python3 bench.py --workload dales_deform --steps 6 --warmup 2 --no-cpu-baseline --distinct-batches 2 > $O/bench_dales_deform_bf16.json 2> $O/bench_dales_deform_bf16.err
echo bench deform done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ks_bench.json 2> /dev/null
cp /tmp/ks/*/*kernel_stats.csv $O/kernel_stats.csv
echo stats done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --prefetch 0 > $O/ks_bench_noprefetch.json 2> /dev/null
cp /tmp/ks2/*/*kernel_stats.csv $O/kernel_stats_noprefetch.csv
echo stats2 done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --prefetch 0 > /dev/null 2>&1
  python3 - /tmp/pm_$c $c >> $O/pmc_per_kernel.csv <<'PY'
import csv,glob,sys,re,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name'])
        n=re.sub(r'\(.*','',n)[:80]
        d[n].append(float(r['Counter_Value']))
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:40]:
    print('%s,"%s",%d,%.1f,%.1f' % (sys.argv[2], n, len(v), sum(v)/len(v), max(v)))
PY
  echo pmc $c done
done
bash tools/train_only_trace.sh > $O/train_only_trace.txt 2>&1
cp gpurun_out/r2_to_agg.csv $O/train_only_kernels.csv
echo train-only trace done
python3 tools/split_gemm_lab.py > $O/split_gemm_lab.txt 2>&1
python3 tools/small_gemm_lab.py > $O/small_gemm_lab.txt 2>&1
echo gemm labs done
