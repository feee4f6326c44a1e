set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${PROFILE_TAG:-v7}
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o pmc --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency-b1 --inflight 1 > $O/pmc_f.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o pmc --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-latency-b1 --inflight 1 > $O/pmc_w.log 2>&1
echo "write pass done"
python3 tools/pmc_summary.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/r01_${PROFILE_TAG:-v7}_pmc_traffic.json > $O/pmc_summary.log
cp $O/r01_${PROFILE_TAG:-v7}_pmc_traffic.json profiles/r01_${PROFILE_TAG:-v7}_pmc_traffic.json
rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 bench.py > $O/r01_${PROFILE_TAG:-v7}_bench_line.json 2> $O/stats.log
echo "stats run done"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r01_${PROFILE_TAG:-v7}_bench_kernel_stats.csv
python3 bench.py --no-cpu-baseline --no-latency-b1 --inflight 1 > $O/r01_${PROFILE_TAG:-v7}_bench_inflight1_line.json 2>> $O/stats.log
python3 bench.py --no-cpu-baseline --steps 20 > $O/r01_${PROFILE_TAG:-v7}_bench_latency_b1_line.json 2>> $O/stats.log
head -12 $O/r01_${PROFILE_TAG:-v7}_bench_kernel_stats.csv
cat $O/pmc_summary.log
