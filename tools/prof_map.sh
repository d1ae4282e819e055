#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel stats of the map-update probe.   tools/prof_map.sh <tag> [particles]
TAG=${1:-x}
NP=${2:-1024}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE_KERNELS=${PROBE_KERNELS:-auto} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $ROOT/tools/probe_fan.py $NP > $OUT/run.log 2>&1
cd $ROOT
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cut -c1-160 "$F" | head -14; else echo "no stats file"; tail -5 $OUT/run.log; fi
