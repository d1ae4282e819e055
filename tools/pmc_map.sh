#!/bin/bash
# Developer tool (GPU box): SQ counter passes of the map-update probe.   tools/pmc_map.sh <tag> [particles]
TAG=${1:-x}
NP=${2:-1024}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE_KERNELS=${PROBE_KERNELS:-auto} timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p1 -o sq -- python3 $ROOT/tools/probe_fan.py $NP > $OUT/p1.log 2>&1
PROBE_KERNELS=${PROBE_KERNELS:-auto} timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -o sq -- python3 $ROOT/tools/probe_fan.py $NP > $OUT/p2.log 2>&1
cd $ROOT
FILES=$(find $OUT -name "*counter_collection.csv")
if [ -n "$FILES" ]; then python3 tools/pmc_sq_collect.py $OUT/sq.json $FILES && python3 -c "
import json; d=json.load(open('$OUT/sq.json'))
for k,v in d.items():
    if 'map_update' in k and v.get('SQ_WAVES',0)>100:
        w=v['SQ_WAVES']; print(k, 'waves',w, 'VALU/wave',round(v['SQ_INSTS_VALU']/w), 'SALU/wave',round(v['SQ_INSTS_SALU']/w), 'LDS/wave',round(v['SQ_INSTS_LDS']/w), 'cycles/wave',round(v.get('SQ_WAVE_CYCLES',0)/w), 'valu_active/wavecyc',round(v.get('SQ_ACTIVE_INST_VALU',0)/max(v.get('SQ_WAVE_CYCLES',1),1),3), 'wait_any',round(v.get('SQ_WAIT_INST_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1),3), 'lds_conf',round(v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1),3))
"; else echo "no counter files"; tail -3 $OUT/p1.log; fi
