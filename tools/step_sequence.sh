# tools/step_sequence.sh <workload> [extra bench args] -- on the GPU box: the ordered kernel sequence of ONE step (one stream), from
# a rocprofv3 kernel trace of bench.py --prefetch 0.  Output: gpurun_out/seq_<workload>.txt (name, duration us, gap to previous us)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; W=${1:-vaihingen}; shift
cd $R; mkdir -p gpurun_out; rm -rf /tmp/seq_$W
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq_$W -- python3 bench.py --workload $W --prefetch 0 --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/seq_${W}_bench.json 2> /dev/null
python3 - /tmp/seq_$W gpurun_out/seq_$W.txt <<'PY'
import csv, glob, sys, re
rows = []
for f in glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the last step: from the last-but-one 'sgd_step_kernel' to the last one
idx = [i for i, n in enumerate(names) if 'sgd_step' in n]
a, b = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
out = open(sys.argv[2], 'w')
prev = None
tot = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n)[:100]
    out.write('%-100s %8.1f %8.1f  grid %s wg %s\n' % (n, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get('Grid_Size', ''), r.get('Workgroup_Size', '')))
    tot += e - s
    prev = e
out.write('# %d launches, %.3f ms of kernels, %.3f ms first start to last end\n' % (b - a, tot / 1e6, (int(rows[b-1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6))
PY
tail -1 gpurun_out/seq_$W.txt
