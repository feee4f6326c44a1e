# Counter passes over the default bench pattern (three batches in flight) and over one batch in flight:
# per-kernel SQ counters (issue, wait, LDS, matrix pipe).  NOTE: counter collection serialises the dispatches, so
# both modes give per-kernel figures without overlap; what the counters cannot show is the co-running regime.
#   bash tools/pmc_overlap.sh   (on the GPU box, via gpurun); summaries: gpurun_out/pmc_overlap/m*_p*.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_overlap
mkdir -p $O
for mode in 3 1; do
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $O/m${mode}_p$i -o pmc --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency-b1 --inflight $mode > $O/m${mode}_p$i.log 2>&1
    echo "mode $mode pass $i done"
    python3 tools/pmc_avg.py $(find $O/m${mode}_p$i -name "*counter_collection.csv" | head -1) > $O/m${mode}_p$i.txt
  done
done
