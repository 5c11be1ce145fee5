# tools/k3_counters.sh -- run on the GPU box: PMC passes over the K3 forward forms on the enc1 layer (tools/kpconv_lab.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-k3c}
mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/k3c_$i -- python3 tools/kpconv_lab.py 2 ${2:-enc1} > $O/run_$i.log 2>&1 || echo "pass $i failed" >> $O/summary.txt
  python3 - /tmp/k3c_$i >> $O/summary.txt <<'PY'
import csv,glob,sys,re,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'kpconv_gather_fwd' not in n: continue
        key='mfma' if 'mfma' in n else 'pool'
        d[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key in d:
    print(key, {c: round(sum(v)/len(v),1) for c,v in d[key].items()}, 'launches', max(len(v) for v in d[key].values()))
PY
  echo "pass $i done" 
done
cat $O/summary.txt
