#!/usr/bin/env python3
"""Instruction mix of one kernel from hipcc's assembly listing, basic block by basic block.

    hipcc ... -S --cuda-device-only -o kernels.s rrt_kernels.hip
    python tools/isa_mix.py kernels.s 'render_fused<float, false, true>' [--blocks] [--md]

The listing is cut at labels (.LBBn_m); every instruction is put in one class:
  pk_f32    v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32       (two FP32 results per lane per issue slot)
  fma_f32   v_fma / v_fmac / v_fmaak / v_fmamk / v_mul / v_add / v_sub / v_mad on f32
  minmax    v_min3 / v_max3 / v_min / v_max / v_med3 on f32
  cmp       v_cmp_* (any type)
  select    v_cndmask
  trans     v_rcp / v_rsq / v_sqrt / v_sin / v_cos / v_exp / v_log  (quarter rate)
  int_mul   v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 / v_mul_u32_u24 ... (quarter rate except the 24-bit forms)
  int       other integer VALU (shift, xor, add, and, bitop3, bfe, cvt between int and float ...)
  mov       v_mov_b32 / v_accvgpr moves
  lane      v_readlane / v_readfirstlane / v_writelane / v_mbcnt / ds_bpermute / DPP moves
  salu      s_* arithmetic, moves, compares, exec-mask handling
  branch    s_cbranch_* / s_branch
  smem      s_load_* / s_buffer_load_*
  wait      s_waitcnt / s_nop / s_barrier
  vmem      global_* / buffer_* / flat_* / scratch_*
  lds       ds_*
Static counts: how often a block runs per loop trip is a property of the workload (the per-trip figures in
profiles/ weigh the blocks with the branch structure of the source: one closest-hit scan, one hit-processing
section, one draw and one new ray per trip).
"""
import argparse
import collections
import re
import subprocess
import sys

CLASSES = ["pk_f32", "fma_f32", "minmax", "cmp", "select", "trans", "int_mul", "int", "mov", "lane", "salu", "branch",
           "smem", "wait", "vmem", "lds"]
VALU = ["pk_f32", "fma_f32", "minmax", "cmp", "select", "trans", "int_mul", "int", "mov", "lane"]


def classify(op):
    if op.startswith("v_pk_") and op.endswith("f32"):
        return "pk_f32"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "ds_bpermute", "ds_permute", "ds_swizzle")) \
            or op.endswith("_dpp"):
        return "lane"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "select"
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", op):
        return "trans"
    if re.match(r"v_(min3|max3|med3|min|max)_f(32|64)", op):
        return "minmax"
    if re.match(r"v_(fma|fmac|fmaak|fmamk|mul|add|sub|subrev|mad|mac)_(f32|f64|legacy_f32)", op):
        return "fma_f32"
    if re.match(r"v_(mul_lo|mul_hi|mad_u64|mad_i64|mul_u32_u24|mul_i32_i24|mad_u32_u24|mad_i32_i24)", op):
        return "int_mul"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "mov"
    if op.startswith("v_"):
        return "int"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return None


def kernel_body(lines, mangled):
    out, on = [], False
    for ln in lines:
        if ln.startswith(mangled + ":"):
            on = True
            continue
        if on:
            if ln.startswith(".Lfunc_end"):
                break
            out.append(ln)
    return out


def find_symbol(lines, demangled):
    syms = [m.group(1) for ln in lines for m in [re.match(r"^(_Z\w+):", ln)] if m]
    names = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
    for s, n in zip(syms, names):
        if n.startswith(("void rrt::" + demangled + "(", "rrt::" + demangled + "(", demangled + "(")) or demangled == s:
            return s
    raise SystemExit("no kernel named %r; have:\n%s" % (demangled, "\n".join(sorted(set(names)))))


def blocks_of(body):
    blocks, cur = [], ("entry", collections.Counter(), [])
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks.append(cur)
            cur = (m.group(1), collections.Counter(), [])
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", ln)
        if not m:
            continue
        c = classify(m.group(1))
        if c:
            cur[1][c] += 1
            cur[2].append(ln.strip())
    blocks.append(cur)
    return blocks


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("listing")
    ap.add_argument("kernel", help="demangled name without the argument list, e.g. 'render_fused<float, false, true>'")
    ap.add_argument("--blocks", action="store_true", help="one line per basic block")
    ap.add_argument("--range", nargs=2, metavar=("FIRST", "LAST"), help="only blocks FIRST..LAST (labels), e.g. the loop")
    ap.add_argument("--ops", action="store_true", help="opcode histogram of the selected blocks")
    a = ap.parse_args()
    lines = open(a.listing).read().split("\n")
    body = kernel_body(lines, find_symbol(lines, a.kernel))
    blocks = blocks_of(body)
    if a.range:
        names = [b[0] for b in blocks]
        i0, i1 = names.index(a.range[0]), names.index(a.range[1])
        blocks = blocks[i0:i1 + 1]
    total = collections.Counter()
    for _, c, _ in blocks:
        total.update(c)
    if a.blocks:
        print("%-12s %5s | " % ("block", "VALU") + " ".join("%7s" % c for c in CLASSES))
        for name, c, _ in blocks:
            print("%-12s %5d | " % (name, sum(c[k] for k in VALU)) + " ".join("%7d" % c[k] for k in CLASSES))
    print("%-12s %5d | " % ("total", sum(total[k] for k in VALU)) + " ".join("%7d" % total[k] for k in CLASSES))
    print(" " * 21 + " ".join("%7s" % c for c in CLASSES))
    if a.ops:
        ops = collections.Counter()
        for _, _, ins in blocks:
            ops.update(i.split()[0] for i in ins)
        for op, n in ops.most_common():
            print("%5d %s" % (n, op))


if __name__ == "__main__":
    main()
