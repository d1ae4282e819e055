#!/bin/bash
# Developer tool (GPU box): the rocprofv3 passes whose summaries go to profiles/ (run from the repository root).
#   tools/profile_round.sh <tag>      e.g. r02_a   -> gpurun_out/prof_<tag>/...
# Passes: kernel trace + stats of bench.py (default configuration, no extra runs), then FETCH_SIZE and WRITE_SIZE counter
# passes of tools/pmc_run.py at 4096 and 1024 particles (counter passes never share a run with trace domains).
set -e
TAG=${1:-r02_a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --no-target-run --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.log
for P in 4096 1024; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${P}_$C -o pmc -- python3 $ROOT/tools/pmc_run.py $P > /dev/null 2> $OUT/pmc_${P}_$C.log
  done
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq1 -o sq -- python3 $ROOT/tools/pmc_run.py 1024 > /dev/null 2> $OUT/sq1.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -o sq -- python3 $ROOT/tools/pmc_run.py 1024 > /dev/null 2> $OUT/sq2.log
find $OUT -name "*.csv" | head -40
