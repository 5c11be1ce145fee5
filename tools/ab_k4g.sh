for v in 0 256 512 0 256 512; do
WEASAL_K4G_INTERLEAVE=$v python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k4g ilv $v', round(d['ms_per_step'],3))"
done
