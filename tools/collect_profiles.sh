#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  tools/collect_profiles.sh r01
# Collects (1) rocprofv3 kernel-trace + stats of the default bench command, (2) FETCH_SIZE and
# (3) WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md: the TCC block cannot hold both, and
# PMC must not be combined with other trace domains).  Everything lands under gpurun_out/<tag>/;
# tools/parse_profiles.py turns it into the committed summaries under profiles/.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python3 $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/bench_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/bench_write.json 2> $OUT/pmc_write.log
# Metropolis-Hastings driver (row f2): kernel trace + stats of a short run on the 12-leaf dataset, 64 and 4096 chains
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mh_trace -- python3 $ROOT/tools/bench_mh.py --iters 40 --cpu-iters 10 > $OUT/mh_steps.json 2> $OUT/mh_trace.log
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
python3 tools/parse_profiles.py $TAG
