# SQ / LDS / TA counter passes over one batch in flight (what roofline.frac is computed from).
#   TAG=r03 bash tools/pmc_r03.sh   (on the GPU box via gpurun) -> gpurun_out/$TAG/pmc_sq.json
# --pmc passes carry --kernel-trace only (no other trace domain).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r03}
O=gpurun_out/$TAG/pmc
mkdir -p $O
CMD=${PMC_CMD:-"bench.py --plain --steps 4 --warmup 1 --inflight 1"}
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc --output-format csv -- python3 $CMD > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; continue; }
  echo "pass $i done"
done
python3 tools/pmc_sq_summary.py $O gpurun_out/$TAG/pmc_sq.json
