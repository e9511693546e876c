#!/usr/bin/env python3
"""Instruction histogram of one kernel of a gfx950 assembly listing, by issue-cost class.

    hipcc <product flags> --cuda-device-only -S vrenderer_amd/csrc/vr_raster.hip -o /tmp/vr_raster.s
    python tools/isa_histogram.py /tmp/vr_raster.s 'k_rasterILb0ELi32ELi1ELb0E' [--blocks]

Classes and SIMD-cycles per wave64 instruction as CALIBRATED in round 4 (tools/micro/valu_cost.hip, wall time of the
kernel x the clock of the launch; profiles/r04_valu_issue_costs.txt): cheap 2 (32 lanes per clock: the datasheet's
157 TFLOP/s), slow 4, transcendental 8.  A lone wave issues one vector instruction per ~4.75 cycles whatever it is.
--blocks prints the per-basic-block table (label, instructions per class, memory operations), which is how the
resolve's per-pixel bodies are told from the coverage sweeps."""
import collections
import re
import sys

CHEAP = 2.0
SLOW = 4.0
TRANS = 8.0

# cheap class: measured at ~2 cycles with eight waves per SIMD
CHEAP_OPS = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_and_b32", "v_or_b32",
             "v_xor_b32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32",
             "v_addc_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
TRANS_OPS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_rcp_f64", "v_rsq_f64"}


def classify(op, operands):
    base = op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if base.startswith(("s_", "buffer_", "global_", "flat_", "ds_", "scratch_")) or base.startswith("v_readlane") or base.startswith("v_readfirstlane"):
        if base.startswith("s_"):
            return "salu" if not base.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_setprio", "s_sleep", "s_cbranch", "s_branch")) else "sctl"
        if base.startswith("ds_"):
            return "lds"
        if base.startswith(("v_readlane", "v_readfirstlane")):
            return "slow"
        return "vmem_st" if "store" in base else "vmem_ld"
    if not base.startswith("v_"):
        return "other"
    if base in TRANS_OPS:
        return "trans"
    if base in CHEAP_OPS:
        # an SGPR or literal operand puts a cheap instruction into the slow class (measured: v_mul_f32 with an sgpr source 4 cycles,
        # with a literal 2.3); inline constants stay cheap
        if re.search(r"(?<![a-z_\[])s\[?\d|\bvcc\b|\bexec\b|\bm0\b", operands) and base not in ("v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subrev_co_u32"):
            return "slow"
        return "cheap"
    return "slow"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9$.]+:", l) and needle in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # the function may hold several s_endpgm (early exits): run to .Lfunc_end
    end = next((i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")), end)
    blocks = []
    cur = ["entry", collections.Counter(), collections.Counter()]
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur)
            cur = [m.group(1), collections.Counter(), collections.Counter()]
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        parts = t.split(None, 1)
        op, operands = parts[0], (parts[1] if len(parts) > 1 else "")
        c = classify(op, operands)
        cur[1][c] += 1
        cur[2][op.replace("_e32", "").replace("_e64", "")] += 1
    blocks.append(cur)
    total = collections.Counter()
    ops = collections.Counter()
    for _, c, o in blocks:
        total.update(c)
        ops.update(o)
    valu = total["cheap"] + total["slow"] + total["trans"]
    cyc = total["cheap"] * CHEAP + total["slow"] * SLOW + total["trans"] * TRANS
    print(f"{needle}: {sum(total.values())} instructions in {len(blocks)} basic blocks (static)")
    print("  " + "  ".join(f"{k}={v}" for k, v in sorted(total.items())))
    print(f"  vector: {valu} = cheap {total['cheap']} + slow {total['slow']} + transcendental {total['trans']}; "
          f"static mean pipe cost {cyc / max(valu, 1):.2f} cycles per instruction")
    print("  most frequent vector opcodes:", ", ".join(f"{k} {v}" for k, v in ops.most_common(40) if k.startswith("v_")))
    if show_blocks:
        print("  per block: label  cheap slow trans | salu lds vmem_ld vmem_st")
        for name, c, _ in blocks:
            n = c["cheap"] + c["slow"] + c["trans"]
            if n + c["vmem_ld"] + c["vmem_st"] + c["lds"] == 0:
                continue
            print(f"    {name:12s} {c['cheap']:4d} {c['slow']:4d} {c['trans']:3d} | {c['salu']:4d} {c['lds']:3d} {c['vmem_ld']:3d} {c['vmem_st']:3d}")


if __name__ == "__main__":
    main()
