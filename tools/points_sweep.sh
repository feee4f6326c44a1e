# frames/s against the points per frame (B=64, two in flight, upload included): the voxeliser's three paths
for n in 4096 8192 16384 32768 65536; do
  python bench.py --plain --points $n --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=%6d  %8.0f frames/s  %.4f ms/step  pillars/frame %.0f' % ($n, d['value'], d['ms_per_step'], d['config'].get('mean_pillars_per_frame', -1)))"
done
