#!/bin/bash
# One bench step under rocprofv3: kernel trace + four separate --pmc passes (never combined with a trace domain).
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh <tag> [bench args...]
# writes gpurun_out/<tag>/{trace,SQ_WAVE_CYCLES,SQ_WAIT_ANY,FETCH_SIZE,WRITE_SIZE}; summarise with tools/pmc_summary.py.
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end "$@" > "$out/trace.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$out/SQ_WAVE_CYCLES" -- python3 bench.py $args > "$out/SQ_WAVE_CYCLES.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$out/SQ_WAIT_ANY" -- python3 bench.py $args > "$out/SQ_WAIT_ANY.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/FETCH_SIZE" -- python3 bench.py $args > "$out/FETCH_SIZE.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE --output-format csv -d "$out/WRITE_SIZE" -- python3 bench.py $args > "$out/WRITE_SIZE.log" 2>&1
ls "$out"
