# SQ counter passes over the K3 forward on the enc1 layer, full and with every memory access ablated (tools/kpconv_lab.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-k3c2}
mkdir -p $O
for abl in 0 15; do
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  WEASAL_K3_ABLATE=$abl rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/k3d_${abl}_$i -- python3 tools/kpconv_lab.py 2 enc1 > $O/run_${abl}_$i.log 2>&1 || echo "pass $i failed" >> $O/summary.txt
  python3 - /tmp/k3d_${abl}_$i $abl >> $O/summary.txt <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'kpconv_gather_fwd' not in n: continue
        key='mfma' if 'mfma' in n else 'pool'
        d[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key in d:
    print('ablate', sys.argv[2], key, {c: round(sum(v)/len(v),1) for c,v in d[key].items()})
PY
  echo "pass $abl $i done"
done
done
cat $O/summary.txt
