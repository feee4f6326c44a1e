# SQ counter passes (issue / wait / LDS / matrix pipe) over one batch in flight, round-2 build.
#   bash tools/pmc_r02.sh   (on the GPU box via gpurun); summaries: gpurun_out/pmc_r02/p*.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r02
mkdir -p $O
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc --output-format csv -- python3 bench.py --plain --steps 4 --warmup 1 --inflight 1 > $O/p$i.log 2>&1
  python3 tools/pmc_avg.py $(find $O/p$i -name "*counter_collection.csv" | head -1) > $O/p$i.txt
  echo "pass $i done"
done
grep -h "k_sep_u<128, 1\|k_sep_u<64, 1\|k_deconv_u\|k_pfn" $O/p*.txt
