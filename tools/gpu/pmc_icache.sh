# instruction-cache counters of the bench kernel: tools/gpu/pmc_icache.sh "<bench args>" <tag>
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT}
OUT=$ROOT/gpurun_out/$2
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-mh --no-graph $1 > $OUT/bench.json 2> $OUT/pmc.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "mcd::" not in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
