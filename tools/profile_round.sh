#!/bin/bash
# Runs ON the GPU box (gpurun): rocprofv3 kernel statistics + the two HBM-traffic counter passes of the bench's default workload, reduced to
# the small files that are committed under profiles/rNN/.  usage: bash tools/profile_round.sh <out dir under gpurun_out/>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( while true; do date >> $O/heartbeat.txt; sleep 45; done ) &      # counter passes are slow and silent: keep the run visibly alive
HB=$!
trap "kill $HB 2>/dev/null" EXIT
MODE=${2:-all}
COMMON="--no-cpu-baseline --no-operating-points --no-trajectory-parity"
if [ "$MODE" != "pmc" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 4 --warmup 1 $COMMON > $O/stats_bench.json 2> $O/stats_bench.log
ST=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp "$ST" $O/kernel_stats.csv
python3 $R/tools/kstats.py $O/kernel_stats.csv 40 > $O/kernel_stats_top.txt 2>&1 || true
rm -rf $O/stats
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 $COMMON --no-roofline > $O/fetch_bench.json 2> $O/fetch_bench.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 $COMMON --no-roofline > $O/write_bench.json 2> $O/write_bench.log
(cd $R && python3 tools/pmc_reduce.py traffic $O/fetch $O/write $O/pmc_traffic.json)
rm -rf $O/fetch $O/write
echo done; ls -la $O
