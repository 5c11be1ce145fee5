# A/B of step-time levers in separate processes (each ~25 s): prints ms/step
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'issue', round(d['host_issue_ms_per_step'],2), 'busy', round(d['host_busy_ms_per_step'],2), 'wait_pf', round(d['host_wait_prefetch_ms_per_step'],2))"; }
WEASAL_PREFETCH_PRIORITY=0 run "default-priority pyramid stream      "
run "high-priority pyramid stream         "
WEASAL_FUSED_MIN_ROWS=0 run "high-priority + all layers block calls"
WEASAL_FUSED_BLOCKS=0 run "high-priority + operator path only    "
WEASAL_PREFETCH_PRIORITY=0 run "default-priority pyramid stream (2)  "
run "high-priority pyramid stream (2)     "
