#!/bin/bash
# ONE GPU job for the numbers a round is judged on: the bench line un-profiled, the same command under
# rocprofv3 --kernel-trace --stats, the four PMC passes over one step, BASELINE config 2's line -- same box, back to back.
# usage (GPU box, repo root): bash tools/final_round.sh r06      -> gpurun_out/<tag>_final/*
set -u
tag=$1
out=gpurun_out/${tag}_final
mkdir -p "$out"
export TMPDIR=/tmp
echo "[final] bench (un-profiled)"; date
python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err" || echo "bench failed"
echo "[final] bench under rocprofv3 --kernel-trace --stats"; date
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-standalone --no-dropin > "$out/trace_bench.json" 2> "$out/trace.err" || echo "trace failed"
echo "[final] PMC passes"; date
args="--steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-standalone --no-dropin"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$out/SQ_WAVE_CYCLES" -- python3 bench.py $args > "$out/pmc1.log" 2>&1
echo "[final] pass 2"; date
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$out/SQ_WAIT_ANY" -- python3 bench.py $args > "$out/pmc2.log" 2>&1
echo "[final] pass 3"; date
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/FETCH_SIZE" -- python3 bench.py $args > "$out/pmc3.log" 2>&1
echo "[final] pass 4"; date
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE --output-format csv -d "$out/WRITE_SIZE" -- python3 bench.py $args > "$out/pmc4.log" 2>&1
echo "[final] config 2"; date
python3 bench.py --indiv 1000 --snps 650000 --steps 20 --warmup 5 --no-end-to-end --no-standalone > "$out/c2_bench.json" 2> "$out/c2.err" || echo "c2 failed"
echo "[final] product-set kernels alone"; date
python3 tools/pw_only.py 5000 1000000 0 > "$out/pw_only.log" 2>&1
ls "$out"
# the trace's stats file is what profiles/<tag>_rocprofv3_kernel_stats.csv is a copy of
find "$out/trace" -name "*kernel_stats.csv" | head -2
