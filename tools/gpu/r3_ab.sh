#!/bin/bash
# same-box A/B/C of builds of the library on the Metropolis-Hastings paths: A = tools/microbench/libtri.so, C = libsix.so (when there), B = the tree's
mkdir -p gpurun_out/r03
for rep in 1 2; do
for lib in A C B; do
  if [ $lib = A ]; then export MCD_LIB_PATH=$(pwd)/tools/microbench/libtri.so; elif [ $lib = C ]; then [ -f tools/microbench/libsix.so ] || continue; export MCD_LIB_PATH=$(pwd)/tools/microbench/libsix.so; else unset MCD_LIB_PATH; fi
  for nl in ${SIZES:-513 400 257 136}; do
    echo -n "$lib $nl "; python tools/bench_mh_large.py $nl 512 3000 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['us_per_lockstep'])"
  done
done
done > gpurun_out/r03/x_ab.txt 2>&1
cat gpurun_out/r03/x_ab.txt
