#!/bin/bash
# kernel trace + the sq1 counter pass only (tools/pmc_passes.sh has all seven).  Usage: tools/pmc_sq1.sh <tag> <bench args...>
set -e
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d gpurun_out/pmc_${tag}_kt -o r -- python3 bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/pmc_${tag}_kt.json 2> gpurun_out/pmc_${tag}_kt.err
python3 tools/prof_summary.py gpurun_out/pmc_${tag}_kt/r_results.db > gpurun_out/pmc_${tag}_kt.txt
rm -rf gpurun_out/pmc_${tag}_kt
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc_${tag}_sq1 -o r -- python3 bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/pmc_${tag}_sq1.json 2> gpurun_out/pmc_${tag}_sq1.err
python3 tools/prof_summary.py gpurun_out/pmc_${tag}_sq1/r_results.db --pmc > gpurun_out/pmc_${tag}_sq1.txt
rm -rf gpurun_out/pmc_${tag}_sq1
