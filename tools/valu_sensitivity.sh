#!/bin/bash
# How VALU-bound is the extend launch?  Run on the GPU box with two libraries built beforehand (on the CPU side):
#   make -C path-tracer_amd/csrc clean all EXTRA=-DPT_EXTRA_VALU=32 && cp path-tracer_amd/libpt_amd.so ab/libpt_extra32.so   (likewise 96)
# PT_AMD_LIB selects the library (capi.py).  For each: the bench (3 pools) and the SQ pass of the exclusive 1 x 2 M pool.
# Usage: tools/valu_sensitivity.sh <out-file> <config> <lib>...
set -e
out=$1; cfg=$2; shift 2; : > $out
export TMPDIR=/tmp
for lib in "$@"; do
  export PT_AMD_LIB=$lib
  timeout -k 10 300 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /tmp/vs.json
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/vs_pmc -o r -- python3 bench.py --config $cfg --pools 1 --paths 2000000 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > /dev/null 2>/tmp/vs.err
  python3 tools/prof_summary.py /tmp/vs_pmc/r_results.db --pmc > /tmp/vs_pmc.txt; rm -rf /tmp/vs_pmc
  python3 - "$lib" >> $out <<'PY'
import json, sys
d = json.load(open('/tmp/vs.json'))
rows = [l for l in open('/tmp/vs_pmc.txt') if l.startswith('k_extend_queue<16, 64, false') or l.startswith('k_extend_queue<12, 256, false')]
vals = {}
for l in rows:
    rest = l[72:].split()
    if len(rest) == 4 and not rest[0][0].isdigit(): vals[rest[0]] = float(rest[3])
    elif len(rest) == 4: vals['avg_us'] = float(rest[2])
cyc = vals['GRBM_GUI_ACTIVE'] / 8
print(f"{sys.argv[1]}: 3 pools {d['value']:.0f} Mrays/s | 1 x 2 M exclusive: launch {vals['avg_us']:.1f} us, VALU insts {vals['SQ_INSTS_VALU'] / 1e6:.1f} M, "
      f"VALU busy {vals['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.3f}, wait-on-data share {vals['SQ_WAIT_INST_ANY'] / vals['SQ_WAVE_CYCLES']:.3f}")
PY
done
