# Round-2 profile set (run on the GPU box through gpurun): one regime per kernel-stats file.
#   TAG=r02a bash tools/profile_r02.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r02}
O=gpurun_out/$TAG
mkdir -p $O
# kernel trace, ONE batch in flight (what roofline.frac is computed from), then the default two in flight
rocprofv3 --kernel-trace --stats -d $O/st1 -o b1 --output-format csv -- python3 bench.py --plain --inflight 1 --steps 200 > $O/plain_inflight1_line.json 2> $O/st1.log
cp $(find $O/st1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_inflight1_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/st3 -o b3 --output-format csv -- python3 bench.py --plain --steps 200 > $O/plain_inflight2_line.json 2> $O/st3.log
cp $(find $O/st3 -name "*kernel_stats.csv" | head -1) $O/${TAG}_inflight2_kernel_stats.csv
# the two secondary legs of the bench line, each by itself
rocprofv3 --kernel-trace --stats -d $O/stk -o bk --output-format csv -- python3 bench.py --only cfgk > $O/cfgk_line.json 2> $O/stk.log
cp $(find $O/stk -name "*kernel_stats.csv" | head -1) $O/${TAG}_cfgk_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/stt -o bt --output-format csv -- python3 bench.py --only train > $O/train_line.json 2> $O/stt.log
cp $(find $O/stt -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_kernel_stats.csv
echo "kernel-trace passes done"
# HBM traffic: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), one batch in flight
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o pmc --output-format csv -- python3 bench.py --plain --steps 4 --warmup 1 --inflight 1 > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o pmc --output-format csv -- python3 bench.py --plain --steps 4 --warmup 1 --inflight 1 > $O/pmc_w.log 2>&1
python3 tools/pmc_summary.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_traffic.json > $O/pmc_summary.log
echo "pmc passes done"
head -14 $O/${TAG}_inflight1_kernel_stats.csv
cat $O/pmc_summary.log
