# A/B of step-time levers in separate processes (each ~25 s): prints ms/step
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'issue', round(d['host_issue_ms_per_step'],2), 'busy', round(d['host_busy_ms_per_step'],2), 'wait_pf', round(d['host_wait_prefetch_ms_per_step'],2))"; }
WEASAL_PREFETCH_WORKERS=1 run "1 pyramid builder "
WEASAL_PREFETCH_WORKERS=2 run "2 pyramid builders"
WEASAL_PREFETCH_WORKERS=3 run "3 pyramid builders"
WEASAL_PREFETCH_WORKERS=2 WEASAL_FUSED_MIN_ROWS=0 run "2 builders + all layers block calls"
WEASAL_PREFETCH_WORKERS=1 run "1 pyramid builder (2)"
WEASAL_PREFETCH_WORKERS=2 run "2 pyramid builders (2)"
