#!/usr/bin/env python3
"""Weighted VALU issue-cost estimate of a kernel from hipcc's -save-temps ISA,
using the costs measured by tools/ubench (profiles/r01_valu_ubench.txt):
full-rate ops 1.0, everything else ~1.65, v_ashr_pk_u8_i32 ~3.3.

    python tools/kcost.py <file.s> <mangled-kernel-name-substring>
"""
import collections
import re
import sys

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_mov_b32", "v_fma_f32", "v_fmac_f32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_add_co_u32",
        "v_addc_co_u32", "v_cndmask_b32", "v_not_b32"}


def main(path, name):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % name, l)][0]
    cnt = collections.Counter()
    for l in lines[start:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s+(v_[a-z0-9_]+)", l)
        if m:
            cnt[re.sub(r"_e32$|_e64$", "", m.group(1))] += 1
    cost = n = 0
    for op, c in cnt.most_common():
        w = 1.0 if op in FULL else (3.3 if op.startswith("v_ashr_pk") else 1.65)
        cost += c * w
        n += c
        if c >= 8:
            print("%-26s %5d x%.2f" % (op, c, w))
    print("VALU instructions %d, weighted cost %.0f full-rate units" % (n, cost))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
