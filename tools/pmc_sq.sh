#!/bin/bash
# Two SQ counter passes (instruction counts, busy and wait shares) of bench.py, for A/B of kernel variants.
# Usage: tools/pmc_sq.sh <tag> <bench args...>   -> gpurun_out/pmc_<tag>_{sq1,sq2}.txt
set -e
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
run_pass() {
  name=$1; counters=$2; shift 2
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $counters -d gpurun_out/pmc_${tag}_${name} -o r -- python3 bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/pmc_${tag}_${name}.json 2> gpurun_out/pmc_${tag}_${name}.err
  python3 tools/prof_summary.py gpurun_out/pmc_${tag}_${name}/r_results.db --pmc > gpurun_out/pmc_${tag}_${name}.txt
  rm -rf gpurun_out/pmc_${tag}_${name}
}
run_pass sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "$@"
run_pass sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "$@"
echo done
