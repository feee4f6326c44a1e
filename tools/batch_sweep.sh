# frames/s against the batch size (two batches in flight, upload included): looks for cliffs in the kernel selection
for b in 1 2 4 8 16 32 64 128; do
  python bench.py --plain --batch $b --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%3d  %8.0f frames/s  %.4f ms/step  %.4f ms/frame' % ($b, d['value'], d['ms_per_step'], d['ms_per_step']/$b))"
done
