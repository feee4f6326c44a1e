for B in ${K4_BATCHES:-2 4 8 16}; do
  echo "== B=$B" 
  python tools/layer_bench.py --batch $B --ablate 0,16 | grep -v deconv
done
