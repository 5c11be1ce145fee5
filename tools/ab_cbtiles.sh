# tools/ab_cbtiles.sh -- on the GPU box (a scratch copy of the repo): rebuilds the library with several CB_TILES / CF_TILES (contrast
# backward / forward: point tiles per wave) and reads the kernels' durations inside a step
run() {
  make -C weasal_amd/csrc > /dev/null 2>&1
  bash tools/step_sequence.sh dales > /dev/null 2>&1
  python3 - "$1" <<'PY'
import re, sys
out = []
for l in open('gpurun_out/seq_dales.txt'):
    if 'contrast_bwd_mfma' in l or 'contrast_reduce2' in l or 'contrast_fwd_mfma' in l:
        m = re.match(r'(.*?)\s+(-?[\d.]+)\s+(-?[\d.]+)\s+grid', l)
        out.append('%s %.1f us' % (m.group(1).replace('void ', '')[:24], float(m.group(2))))
print(sys.argv[1], '; '.join(out))
PY
}
for v in 2 4; do
  sed -i "s/^constexpr int CB_TILES = [0-9]*;/constexpr int CB_TILES = $v;/" weasal_amd/csrc/contrast_mfma.hip
  run "CB_TILES=$v CF_TILES=8:"
done
for v in 2 4; do
  sed -i "s/^constexpr int CF_TILES = [0-9]*;/constexpr int CF_TILES = $v;/" weasal_amd/csrc/contrast_mfma.hip
  run "CB_TILES=4 CF_TILES=$v:"
done
