# A/B of step-time levers in separate processes (each ~25 s): prints ms/step.  Usage: ab_bench.sh "<label>=<env assignments>" ...
# BENCH_ARGS: extra bench.py arguments for every run (e.g. "--prefetch 0")
cd $GRAFT_REPO_ROOT
run() { env $2 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 $BENCH_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'issue', round(d['host_issue_ms_per_step'],2), 'busy', round(d['host_busy_ms_per_step'],2), 'wait_pf', round(d['host_wait_prefetch_ms_per_step'],2), 'wait_gpu', round(d['host_wait_gpu_ms_per_step'],2))"; }
for rep in 1 2 3; do
  for spec in "$@"; do
    run "${spec%%=*}" "${spec#*=}"
  done
done
