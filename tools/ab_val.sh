# tools/ab_val.sh <ENV VAR> <value a> <value b> [workload] -- bench.py with VAR=a and VAR=b alternating three times
V=$1; A=$2; B=$3; W=${4:-dales}
for rep in 1 2 3; do
  for x in $A $B; do
    env $V=$x python3 bench.py --workload $W --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $V=$x', round(d['ms_per_step'],3))"
  done
done
