#!/usr/bin/env python3
"""Issue-cost estimate per basic block of one kernel in a gfx950 assembly listing
(hipcc -S --cuda-device-only): counts vector, scalar, LDS and memory
instructions and prices the vector ones at the issue rates measured by
tools/ubench_valu.hip (2 cycles for the plain 32-bit add/sub/logic/shift-right/
mov family in VOP1/VOP2 form, 4 for the rest).

    python tools/isa_cost.py /tmp/group.s k_select_quad [min_instructions]
"""
import collections
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_mov_b32", "v_not_b32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32",
        "v_add_u16", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_write_b32",
        "v_accvgpr_read_b32"}


def cost(op, operands):
    base = op.replace("_e32", "").replace("_e64", "")
    if "_dpp" in op or "_sdwa" in op:
        return 4
    base = base.replace("_dpp", "").replace("_sdwa", "")
    if base in FAST and not op.endswith("_e64"):
        # an SGPR operand (not a literal or inline constant) makes it a half-rate issue
        if re.search(r"\bs\d+\b|\bs\[\d+", operands.split(",", 1)[1] if "," in operands else ""):
            return 4
        return 2
    return 4


def main(path, kernel, floor=8):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kernel in l and "; @" in l)
    blocks = []
    name, body = "entry", []
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, body))
            name, body = m.group(1), []
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        body.append(t.split(";")[0].strip())
    blocks.append((name, body))
    total = collections.Counter()
    for name, body in blocks:
        c = collections.Counter()
        ops = collections.Counter()
        for ins in body:
            parts = ins.split(None, 1)
            op, rest = parts[0], parts[1] if len(parts) > 1 else ""
            if op.startswith("v_"):
                k = cost(op, rest)
                c["valu"] += 1
                c["valu_cycles"] += k
                ops[op] += 1
            elif op.startswith("s_waitcnt"):
                c["wait"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
                c["vmem"] += 1
                if op.startswith("scratch_"):
                    c["scratch"] += 1
        total.update(c)
        if sum(c.values()) - c["valu_cycles"] >= floor:
            top = ", ".join("%s %d" % kv for kv in ops.most_common(6))
            print("%-12s valu %4d (%5d cyc) salu %4d lds %3d vmem %3d (scratch %d) wait %3d | %s" % (
                name, c["valu"], c["valu_cycles"], c["salu"], c["lds"], c["vmem"], c["scratch"], c["wait"], top))
    print("total        valu %4d (%5d cyc) salu %4d lds %3d vmem %3d (scratch %d) wait %3d" % (
        total["valu"], total["valu_cycles"], total["salu"], total["lds"], total["vmem"], total["scratch"],
        total["wait"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
