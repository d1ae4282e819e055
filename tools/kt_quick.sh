#!/bin/bash
# Developer tool (GPU box): kernel trace + stats of one default bench.py run, per-kernel table to stdout.
#   tools/kt_quick.sh [bench.py arguments]
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kt_quick
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py --no-target-run --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/kt.log
F=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
if [ -z "$F" ]; then echo "no kernel_stats.csv"; tail -5 $OUT/kt.log; exit 1; fi
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[:16]:
    print(r[0][:60].ljust(60), *[x[:12].rjust(12) for x in r[1:5]])
PY
