#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (one counter block budget per pass, no tracing
# domains besides --kernel-trace).  Usage: tools/pmc_passes.sh <tag> <bench args...>   -> gpurun_out/pmc_<tag>_<pass>/
# Each pass runs under its own timeout; a failing pass stops the script.
set -e
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
run_pass() {
  name=$1; counters=$2; shift 2
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $counters -d gpurun_out/pmc_${tag}_${name} -o r -- python3 bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/pmc_${tag}_${name}.json 2> gpurun_out/pmc_${tag}_${name}.err
  python3 tools/prof_summary.py gpurun_out/pmc_${tag}_${name}/r_results.db --pmc > gpurun_out/pmc_${tag}_${name}.txt
  rm -rf gpurun_out/pmc_${tag}_${name}     # the trace database is large; gpurun_out only travels back under 64 MiB
}
# first the same command with --kernel-trace only: the launch durations no counter collection has touched
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d gpurun_out/pmc_${tag}_kt -o r -- python3 bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/pmc_${tag}_kt.json 2> gpurun_out/pmc_${tag}_kt.err
python3 tools/prof_summary.py gpurun_out/pmc_${tag}_kt/r_results.db > gpurun_out/pmc_${tag}_kt.txt
rm -rf gpurun_out/pmc_${tag}_kt
run_pass sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "$@"
run_pass sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "$@"
run_pass tcc1 "FETCH_SIZE TCC_HIT_sum" "$@"
run_pass tcc2 "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "$@"
run_pass ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" "$@"
run_pass tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "$@"
echo done
