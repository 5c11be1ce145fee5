for v in 0 256 512 0 256 512; do
WEASAL_POOL_INTERLEAVE=$v python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ilv $v', round(d['ms_per_step'],3))"
done
