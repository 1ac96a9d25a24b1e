# round 4, first GPU pass: bench.py started bare with --gpus 2 (self-launched ranks, one-device rehearsal), then SQ counter
# passes on the lock-step Metropolis-Hastings run (257 and 1025 nodes x 512 chains) -> what the waves of a lock step wait for.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04/pmc
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -q -m gpu -x > gpurun_out/r04/a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04/a_tests.log
tail -15 gpurun_out/r04/a_tests.log
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/pmc
cd /tmp
MH257="python3 $GRAFT_REPO_ROOT/bench.py --kind mh --steps 8000 --warmup 800"
MH1025="python3 $GRAFT_REPO_ROOT/bench.py --kind mh --dim 1024 --chains 512 --steps 4000 --warmup 400"
for tag in mh257 mh1025; do
  if [ $tag = mh257 ]; then CMD="$MH257"; else CMD="$MH1025"; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/${tag}_sq1 -- $CMD > $OUT/${tag}_sq1.json 2> $OUT/${tag}_sq1.log && echo "$tag sq1 ok"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/${tag}_sq2 -- $CMD > $OUT/${tag}_sq2.json 2> $OUT/${tag}_sq2.log && echo "$tag sq2 ok"
done
ls -R $OUT | head -40
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/microbench/headline_phases.py 256 512 > gpurun_out/r04/headline_phases.txt 2> gpurun_out/r04/headline_phases.err; echo "phases rc=$?"
cat gpurun_out/r04/headline_phases.txt
