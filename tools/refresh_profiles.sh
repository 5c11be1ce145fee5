# tools/refresh_profiles.sh -- on the GPU box: every bench line and kernel-stats summary that profiles/README.md lists for the
# round, written under gpurun_out/refresh/ (copy what is to be judged into profiles/ afterwards).  ~6 minutes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/refresh; mkdir -p $O; rm -rf /tmp/rp_*
line() { tail -1; }
python3 bench.py 2>/dev/null | line > $O/bench_dales_1gpu.json && echo default done
python3 bench.py --contrast 0 --no-cpu-baseline 2>/dev/null | line > $O/bench_dales_no_contrast_1gpu.json
python3 bench.py --workload vaihingen --no-cpu-baseline 2>/dev/null | line > $O/bench_vaihingen_1gpu.json
python3 bench.py --workload vaihingen_wl --no-cpu-baseline 2>/dev/null | line > $O/bench_vaihingen_wl_1gpu.json
python3 bench.py --workload dales_deform --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line > $O/bench_dales_deform_bf16_1gpu.json && echo deform done
python3 bench.py --nearest-upsample 1 --no-cpu-baseline 2>/dev/null | line > $O/bench_dales_nearest_upsample_optin_1gpu.json
python3 bench.py --nearest-upsample 1 --workload dales_deform --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line > $O/bench_dales_deform_bf16_nearest_upsample_optin_1gpu.json
python3 bench.py --mode infer --steps 30 --no-cpu-baseline 2>/dev/null | line > $O/bench_dales_infer_1gpu.json
python3 bench.py --mode infer --steps 30 --prefetch 0 --no-cpu-baseline 2>/dev/null | line > $O/bench_dales_infer_noprefetch_1gpu.json && echo infer done
python3 tools/train_only_bench.py 40 contrast 2>&1 | tail -1 > $O/train_only.txt
python3 tools/train_only_bench.py 40 2>&1 | tail -1 >> $O/train_only.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_default -- python3 bench.py > $O/bench_dales_1gpu_under_rocprofv3.json 2> /dev/null
cp /tmp/rp_default/*/*kernel_stats.csv $O/kernel_stats_bench_default_command.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_5 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/rp_5/*/*kernel_stats.csv $O/kernel_stats_dales_5steps.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_5n -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --prefetch 0 > /dev/null 2>&1
cp /tmp/rp_5n/*/*kernel_stats.csv $O/kernel_stats_dales_5steps_noprefetch.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_d -- python3 bench.py --workload dales_deform --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/rp_d/*/*kernel_stats.csv $O/kernel_stats_dales_deform_4steps.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_v -- python3 bench.py --workload vaihingen --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/rp_v/*/*kernel_stats.csv $O/kernel_stats_vaihingen_20steps.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_i -- python3 bench.py --mode infer --steps 10 --warmup 2 --prefetch 0 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/rp_i/*/*kernel_stats.csv $O/kernel_stats_dales_infer_10steps_noprefetch.csv
ls $O | wc -l
