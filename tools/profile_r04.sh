# Round-4 profile set (run on the GPU box through gpurun): one regime per kernel-stats file.
#   PP_GIT_HEAD=<short hash> TAG=r04 bash tools/profile_r04.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r04}
O=gpurun_out/$TAG
mkdir -p $O
kstats() {   # kstats <dir> <out csv> -- <python args...>
  local d=$1 out=$2; shift 3
  rocprofv3 --kernel-trace --stats -d $O/$d -o t --output-format csv -- python3 "$@" > $O/$d.json 2> $O/$d.log
  cp $(find $O/$d -name "*kernel_stats.csv" | head -1) $O/$out
}
# kernel trace, ONE batch in flight (what roofline.frac is computed from), then the default two in flight
kstats st1 ${TAG}_inflight1_kernel_stats.csv -- bench.py --plain --inflight 1 --steps 200
kstats st2 ${TAG}_inflight2_kernel_stats.csv -- bench.py --plain --steps 200
# the secondary legs, each by itself: cfg-K (one batch in flight = one regime), training at B=2 and B=32
kstats stk ${TAG}_cfgk_inflight1_kernel_stats.csv -- bench.py --only cfgk --inflight 1
kstats stt ${TAG}_train_b2_kernel_stats.csv -- bench.py --only train
kstats stt32 ${TAG}_train_b32_kernel_stats.csv -- bench.py --only train --train-batch 32
kstats stt64 ${TAG}_train_b64_kernel_stats.csv -- bench.py --only train --train-batch 64
echo "kernel-trace passes done"
# HBM traffic: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), one batch in flight; cfg-A then cfg-K
for leg in a k; do
  if [ $leg = a ]; then CMD="bench.py --plain --steps 4 --warmup 1 --inflight 1"; else CMD="bench.py --only cfgk --inflight 1 --steps 4"; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f$leg -o pmc --output-format csv -- python3 $CMD > $O/pmc_f$leg.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w$leg -o pmc --output-format csv -- python3 $CMD > $O/pmc_w$leg.log 2>&1
done
python3 tools/pmc_summary.py $(find $O/pmc_fa -name "*counter_collection.csv" | head -1) $(find $O/pmc_wa -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_traffic.json > $O/pmc_summary.log
python3 tools/pmc_summary.py $(find $O/pmc_fk -name "*counter_collection.csv" | head -1) $(find $O/pmc_wk -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_traffic_cfgk.json > $O/pmc_summary_k.log
echo "traffic passes done"
TAG=$TAG bash tools/pmc_r03.sh > $O/pmc_sq.log 2>&1
cp $O/pmc_sq.json $O/${TAG}_pmc_sq.json
echo "SQ passes done"
head -16 $O/${TAG}_inflight1_kernel_stats.csv
cat $O/pmc_summary.log
tail -14 $O/pmc_sq.log
