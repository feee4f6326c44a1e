# Tuning aid: split-K kernels (a0) against the uniform-wave kernels (a16 forces them) per layer and batch size.
#   K4_BATCHES="2 4 8" PP_SEP_K4=8 PP_DECONV_K4=8 bash tools/k4_sweep.sh
for B in ${K4_BATCHES:-2 4 8 16}; do
  echo "== B=$B"
  python tools/layer_bench.py --batch $B --ablate 0,16
done
