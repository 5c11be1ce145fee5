# tools/ab.sh <env var> <workload...> -- on the GPU box: bench.py with VAR=0 and VAR=1, alternating twice per workload
V=$1; shift
mkdir -p gpurun_out
for w in "$@"; do
  for rep in 1 2; do
    for x in 0 1; do
      env $V=$x python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $V=$x', round(d['ms_per_step'],3), 'host_busy', round(d['host_busy_ms_per_step'],2), 'wait_prefetch', round(d['host_wait_prefetch_ms_per_step'],2))"
    done
  done
done
