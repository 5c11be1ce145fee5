# tools/capture_default_stats.sh -- on the GPU box: rocprofv3 kernel stats of EXACTLY the default benchmark command (python3 bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/default_stats; rm -rf /tmp/ks_default
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_default -- python3 bench.py > gpurun_out/default_stats/bench_under_rocprof.json 2> /dev/null
cp /tmp/ks_default/*/*kernel_stats.csv gpurun_out/default_stats/kernel_stats_bench_default.csv
head -4 gpurun_out/default_stats/kernel_stats_bench_default.csv | cut -c1-200
