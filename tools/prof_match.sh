#!/bin/bash
# Matcher diagnostics on the GPU box: phase stamps of match_kernel and ndt_kernel (stamped build, then a normal rebuild).
# usage: tools/prof_match.sh [P]
set -e
P=${1:-4096}
mkdir -p gpurun_out
RBPF_STAMPS=match python -m thesis_amd.build --force > gpurun_out/pm_build.log 2>&1
timeout -k 10 200 python tools/probe_match_stamps.py $P > gpurun_out/pm_match_stamps.txt 2>&1
timeout -k 10 200 python tools/probe_ndt_stamps.py $P > gpurun_out/pm_ndt_stamps.txt 2>&1
python -m thesis_amd.build --force >> gpurun_out/pm_build.log 2>&1
cat gpurun_out/pm_match_stamps.txt gpurun_out/pm_ndt_stamps.txt
