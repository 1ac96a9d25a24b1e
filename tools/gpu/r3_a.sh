# round 3, first GPU pass: new tests (MC3 swap on the device, library mass adaptation), the driver's default line, config 5 as one command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 500 python -m pytest tests/test_gpu_mh.py tests/test_gpu_nuts.py -q -m gpu -x -k "mc3 or nuts" > gpurun_out/r03/a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03/a_tests.log
tail -5 gpurun_out/r03/a_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/a_bench_default.json 2> gpurun_out/r03/a_bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --kind mh --n 1024 --chains 512 --swap-period 2 --swap-steps 500 --steps 2000 --warmup 200 > gpurun_out/r03/a_bench_cfg5_1gpu.json 2> gpurun_out/r03/a_bench_cfg5_1gpu.err; echo "cfg5 rc=$?"
MCD_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --kind mh --n 256 --chains 64 --steps 600 --warmup 100 --swap-steps 200 > gpurun_out/r03/a_bench_cfg5_rehearsal.json 2> gpurun_out/r03/a_bench_cfg5_rehearsal.err; echo "rehearsal rc=$?"
tail -c 1500 gpurun_out/r03/a_bench_default.json; tail -c 600 gpurun_out/r03/a_bench_cfg5_1gpu.json; tail -c 600 gpurun_out/r03/a_bench_cfg5_rehearsal.json
