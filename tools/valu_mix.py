#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel's main loop, priced with the measured issue costs of gfx950
(profiles/r01_ubench_issue_rates_gfx950.txt: cycles one wave64 instruction occupies its SIMD with 8 waves resident).

    hipcc --offload-arch=gfx950 <the Makefile's FLAGS> -S --cuda-device-only pt_kernels_extend.hip -o /tmp/ext.s
    tools/valu_mix.py /tmp/ext.s '<mangled-name substring>'      e.g. k_extend_queueILi16ELi64ELb0ELi2ELb0ELb1ELb0ELi0ELb0ELb1E

Prints the mix of the longest depth-1 loop of the kernel (the traversal loop) and the mean cycles per VALU instruction, the
factor bench.py's `valu` ceiling uses instead of the nominal 4 (SQ_ACTIVE_INST_VALU x 4 exceeds the SIMD cycles when
instructions are added: profiles/r02_valu_sensitivity_sphere.txt)."""
import collections
import re
import sys

COST = {"transcendental (v_rcp/v_sqrt/v_rsq)": 8.3, "min/max/cmp/cndmask/bfi/div_*/3-operand integer": 4.4, "packed f32": 4.6, "plain VOP1/VOP2/VOP3 f32 and integer": 2.5}


def classify(op):
    if op.startswith(("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32")):
        return "transcendental (v_rcp/v_sqrt/v_rsq)"
    if op.startswith(("v_min", "v_max", "v_cmp", "v_bfi", "v_lshl_add", "v_div_scale", "v_div_fmas", "v_div_fixup", "v_cndmask", "v_med3", "v_mad_u64",
                      "v_lshl_or", "v_and_or", "v_add3")):
        return "min/max/cmp/cndmask/bfi/div_*/3-operand integer"
    if op.startswith("v_pk_"):
        return "packed f32"
    return "plain VOP1/VOP2/VOP3 f32 and integer" if op.startswith("v_") else None


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l and l.rstrip().endswith(":") or (l.startswith("_Z") and sys.argv[2] in l and ": " in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    best = None                                                                           # the traversal loop is the kernel's longest depth-1 loop
    for h in (i for i in range(start, end) if "Loop Header: Depth=1" in lines[i]):
        name = lines[h].split(":")[0].strip().lstrip(".L")                                # comments name blocks without the ".L"
        inside = [i for i in range(h, end) if f"Header={name} " in lines[i] or f"Parent Loop {name} " in lines[i]]
        b = max(inside) if inside else h
        while b + 1 < end and not re.match(r"\.LBB", lines[b + 1]):                       # to the end of the loop's last block
            b += 1
        if best is None or b - h > best[1] - best[0]:
            best = (h, b)
    head, back = best
    mix, mem = collections.Counter(), collections.Counter()
    for l in lines[head:back + 1]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m:
            continue
        k = classify(m.group(1))
        if k:
            mix[k] += 1
        elif m.group(1).startswith(("global_", "ds_", "buffer_", "flat_", "scratch_")):
            mem[m.group(1)] += 1
    n = sum(mix.values()); cyc = sum(COST[k] * v for k, v in mix.items())
    print(f"# kernel {lines[start].split(':')[0]}\n# main loop: asm lines {head - start + 1}..{back - start + 1} of the kernel, {n} VALU instructions")
    for k, v in mix.most_common():
        print(f"{v:>6}  x {COST[k]:>4} cycles  {k}")
    print(f"mean cycles per VALU instruction: {cyc / n:.2f}")
    print("memory instructions in the loop:", dict(mem))


if __name__ == "__main__":
    main()
