#!/usr/bin/env python3
"""Turn the PMC passes of tools/pmc_passes.sh into the committed inputs of bench.py's `roofline` block.

    tools/pmc_roofline.py <round-tag> <config> <pools>x<paths> <pass-prefix> [<valu-mix file of tools/valu_mix.py>]
        e.g.  tools/pmc_roofline.py r02 sphere 1x2000000 gpurun_out/pmc_r2_sphere_p1 profiles/r02_valu_mix_sphere.txt

copies <pass-prefix>_{kt,sq1,sq2,tcc1,tcc2,ea,tcp}.txt to profiles/<round-tag>_pmc_<config>_<shape>_<pass>.txt and writes
profiles/roofline_<config>_<shape>.json: per extend launch (the timed, non-counting variant of k_extend_queue) the kernel's own
duration, HBM-side bytes, VALU instructions and busy share, cache hit rates and wave occupancy -- each with the file it
was read from.  Formulas:
  HBM read bytes   = 32*RDREQ_32B + 128*RDREQ_128B + 64*(RDREQ - RDREQ_32B - RDREQ_128B)      (TCC_EA0_RDREQ*: exact per size;
                     FETCH_SIZE tallies 128-B requests at 64 B on gfx950, MI355X_MICROARCH.md "HBM")
  HBM write bytes  = WRITE_SIZE (KiB), exact for 16-B streaming stores and float atomics per the same section
  VALU busy        = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)         (SQ_* count quad-cycles, GRBM sums 8 XCDs)
                     -- an upper bound: it passes 1 when instructions are added (profiles/r02_valu_sensitivity_sphere.txt)
  VALU issue busy  = SQ_INSTS_VALU * (mean measured issue cycles of the loop's instruction mix, valu-mix file) / (1024 SIMDs * cycles)
  occupancy        = SQ_WAVE_CYCLES * 4 / (GRBM_GUI_ACTIVE / 8) / 8192 wave slots
  L1 hit           = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES ;  L2 hit = TCC_HIT / TCC_REQ
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, config, shape, prefix = sys.argv[1:5]
mix_file = sys.argv[5] if len(sys.argv) > 5 else None
PASSES = ("kt", "sq1", "sq2", "tcc1", "tcc2", "ea", "tcp")     # kt first: its launch duration (no counters collected) is the one kept


def parse(path):
    """-> {kernel: {"calls", "avg_us", counters...}}; counters are means per dispatch."""
    out = {}
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel") or not line.strip():
            continue
        name = line[:72].strip()
        rest = line[72:].split()
        k = out.setdefault(name, {})
        if len(rest) == 4 and re.match(r"^[0-9.]+$", rest[0]):          # calls total_us avg_us pct
            k["calls"], k["avg_us"] = int(rest[0]), float(rest[2])
        elif len(rest) == 4:                                            # counter n sum mean
            k[rest[0]] = float(rest[3])
    return out


def main():
    vals, src = {}, {}
    for p in PASSES:
        f = f"{prefix}_{p}.txt"
        if p == "kt" and not os.path.exists(f):
            continue
        dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_{config}_{shape}_{p}.txt")
        shutil.copyfile(f, dst)
        tab = parse(f)
        # the render's extend launches: k_extend_queue<..., COUNT = false, ...> with the logic tail (most calls)
        def targs(n):                                                   # <LDS_N, BLOCK, COUNT, ORDER, LOG, LOGIC, ...>
            return [a.strip() for a in n[n.index("<") + 1:].rstrip(">.").split(",")]
        names = [n for n in tab if n.startswith("k_extend_queue<") and targs(n)[2] == "false" and targs(n)[5] == "true"]
        name = max(names, key=lambda n: tab[n].get("calls", 0))
        if p == "sq1":
            vals["sq1_avg_us"] = tab[name]["avg_us"]
        for c, v in tab[name].items():
            vals.setdefault(c, v); src.setdefault(c, os.path.basename(dst))
        vals["kernel"] = name
    rd = 32 * vals["TCC_EA0_RDREQ_32B_sum"] + 128 * vals["TCC_EA0_RDREQ_128B_sum"] + 64 * (vals["TCC_EA0_RDREQ_sum"] - vals["TCC_EA0_RDREQ_32B_sum"] - vals["TCC_EA0_RDREQ_128B_sum"])
    wr = vals["WRITE_SIZE"] * 1024
    cyc = vals["GRBM_GUI_ACTIVE"] / 8
    out = {
        "kernel": vals["kernel"], "config": config, "shape": shape, "launches_profiled": vals["calls"],
        "rocprof_avg_launch_us": vals["avg_us"], "rocprof_avg_launch_us_source": src["avg_us"],
        "hbm_read_bytes_per_launch": round(rd, 1), "hbm_write_bytes_per_launch": round(wr, 1), "hbm_bytes_per_launch": round(rd + wr, 1),
        "valu_insts_per_launch": vals["SQ_INSTS_VALU"], "salu_insts_per_launch": vals["SQ_INSTS_SALU"],
        "valu_busy": round(vals["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), 4),
        "occupancy": round(vals["SQ_WAVE_CYCLES"] * 4 / cyc / 8192, 4),
        "wait_on_data_share_of_wave_time": round(vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"], 4),
        "l1_hit": round(1 - vals["TCP_TCC_READ_REQ_sum"] / vals["TCP_TOTAL_CACHE_ACCESSES_sum"], 4),
        "l2_hit": round(vals["TCC_HIT_sum"] / vals["TCC_REQ_sum"], 4),
        "gpu_cycles_per_launch": round(cyc, 1), "gpu_clock_ghz": round(cyc / (vals["sq1_avg_us"] * 1e3), 4),     # both from the sq1 pass
        "sources": sorted(set(src.values())),
    }
    if mix_file:
        cpi = float(re.search(r"mean cycles per VALU instruction: ([0-9.]+)", open(mix_file).read()).group(1))
        out["valu_cycles_per_inst"] = cpi
        out["valu_issue_busy"] = round(vals["SQ_INSTS_VALU"] * cpi / (1024 * cyc), 4)
        out["sources"].append(os.path.basename(mix_file))
    path = os.path.join(ROOT, "profiles", f"roofline_{config}_{shape}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
