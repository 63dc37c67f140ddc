"""Counts the instructions on the common path of one loop of a gfx950 kernel's assembly
(hipcc -S --cuda-device-only): starts at a loop-header label and follows the path on which every
`s_cbranch_execz` is TAKEN (rare per-lane blocks are skipped) and every other conditional branch
falls through, until the back-edge.  Classes follow the issue rates measured by tools/instr_rate.hip
(profiles/r01_instr_rates.txt): full-rate VALU (VOP1/VOP2 2-operand forms), half-rate VALU
(VOP3 3-operand forms, 32-bit multiplies, v_mad_u64_u32, 64-bit forms, compares and carries through VCC,
v_cndmask, SDWA forms, and any plain form with a scalar-register operand -- profiles/r02_instr_rates.txt), LDS,
SALU, other.
Usage: python tools/isa_count.py kernel.s LOOP_LABEL"""
import re
import sys
from collections import Counter

HALF = re.compile(r"^v_(mul_lo_u32|mul_hi_u32|mad_u64_u32|mad_u32_u24|mul_u32_u24|alignbit_b32|alignbyte_b32|perm_b32|bfe_u32|"
                  r"and_or_b32|add3_u32|lshl_add_u32|lshl_or_b32|or3_b32|xad_u32|lshl_add_u64|lshlrev_b64|lshrrev_b64|"
                  r"xor3_b32|bfi_b32|add_lshl_u32|cndmask_b32|mad_i32_i24|mul_i32_i24|mul_hi_i32|mov_b64|cmp_|add_co_|addc_co_|"
                  r"sub_co_|subb_co_|subrev_co_|bitop3_b32)")
SCALAR_OPERAND = re.compile(r"(^|[ ,\[])(s\d+|s\[\d+:\d+\]|vcc|exec)\b")


def is_half(op, line):
    return bool(HALF.match(op)) or "_sdwa" in op or bool(SCALAR_OPERAND.search(line[len(op):]))


def main():
    path, label = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    i = labels[label] + 1
    cnt, ops = Counter(), Counter()
    steps = 0
    while steps < 100000 and i < len(lines):
        steps += 1
        if lines[i].startswith(label + ":"):      # fell through into the header again: one iteration done
            break
        l = lines[i].strip()
        i += 1
        if not l or l.startswith(";") or l.startswith("."):
            continue
        op = l.split()[0]
        if op == "s_cbranch_execz":
            if l.split()[1] == label:     # a skipped rare block that ends the iteration
                cnt["salu"] += 1
                break
            i = labels[l.split()[1]] + 1
            cnt["salu"] += 1
            continue
        if op == "s_branch":
            tgt = l.split()[1]
            if tgt == label:
                break
            i = labels[tgt] + 1
            cnt["salu"] += 1
            continue
        if op.startswith("s_cbranch"):
            tgt = l.split()[1]
            cnt["salu"] += 1
            if tgt == label:        # back-edge (taken)
                break
            continue
        ops[op] += 1
        if op.startswith("v_"):
            cnt["valu_half" if is_half(op, l) else "valu_full"] += 1
        elif op.startswith("ds_"):
            cnt["lds"] += 1
        elif op.startswith("s_"):
            cnt["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            cnt["vmem"] += 1
        else:
            cnt["other"] += 1
    print(dict(cnt), "valu total", cnt["valu_half"] + cnt["valu_full"], "issue units (full=1, half=2):",
          cnt["valu_full"] + 2 * cnt["valu_half"])
    for op, n in ops.most_common():
        print("%5d  %s" % (n, op))


main()
