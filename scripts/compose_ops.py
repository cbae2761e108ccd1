#!/usr/bin/env python3
"""Instruction mix per step of the rank-10 composition kernel (mtg_tp_big.h, structure NR = 0, NC = 5: configs[4]),
role by role: the four roles of tpb4_compose_eval<0, 5> compiled as four kernels, the steady-state loop of each (the
largest loop of the function) counted by instruction class.  Development aid behind DESIGN.md's table of where the
composition's vector instructions go.   python scripts/compose_ops.py [extra hipcc flags]"""
import collections, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = "/tmp/mtg_compose_ops"
os.makedirs(WORK, exist_ok=True)
SRC = r'''
#define MTG_EXP_BITS 10
#define MTG_TRIG_BITS 9
#include "mtg_tp_big.h"
template <int ROLE>
__global__ void __launch_bounds__(256, 2) role_kernel(MtgSolveArgs a, double *elems, double *parts, int C)
{
    __shared__ TpbRing4<10> ring;
    __shared__ MtgMathTables tab;
    mtg_fill_tables(&tab, threadIdx.x, 256);
    __syncthreads();
    tpb4_compose_eval<0, 5>(a, blockIdx.y, elems, parts, C, ring, &tab, blockIdx.x, ROLE);
}
template __global__ void role_kernel<0>(MtgSolveArgs, double *, double *, int);
template __global__ void role_kernel<1>(MtgSolveArgs, double *, double *, int);
template __global__ void role_kernel<2>(MtgSolveArgs, double *, double *, int);
template __global__ void role_kernel<3>(MtgSolveArgs, double *, double *, int);
'''
open(os.path.join(WORK, "c5.hip"), "w").write(SRC)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "mind_the_gaps_amd", "csrc"),
                       "-I" + os.path.join(ROOT, "include"), *sys.argv[1:], "-S", "--cuda-device-only", "c5.hip", "-o", "c5.s"],
                      cwd=WORK, stderr=subprocess.DEVNULL)
s = open(os.path.join(WORK, "c5.s")).read()
ROLES = ("0 columns, low", "1 columns, high (+ b, z)", "2 filter (+ ch, pivots)", "3 filter")


def classify(op):
    if op.startswith("v_"):
        if "f64" in op:
            return "valu_f64"
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


total = collections.Counter()
for role in range(4):
    name = "_Z11role_kernelILi%dEEv12MtgSolveArgsPdS1_i" % role
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    lines = [l.strip() for l in s[i:j].split("\n")]
    # blocks of every loop: the header's block plus the blocks the compiler marks "in Loop: Header=..."; a block runs from
    # its label to the next label
    labels = [k for k, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)] + [len(lines)]
    loops = collections.defaultdict(list)
    for a_, b_ in zip(labels[:-1], labels[1:]):
        head = lines[a_]
        lab = head.split(":")[0].lstrip(".L")
        m = re.search(r"in Loop: Header=(BB\d+_\d+)", head)
        if "Loop Header" in head:
            loops[lab].append((a_, b_))
        elif m:
            loops[m.group(1)].append((a_, b_))
    body = max(loops.values(), key=lambda blocks: sum(b_ - a_ for a_, b_ in blocks))
    lo, hi = min(a_ for a_, _ in body), max(b_ for _, b_ in body)
    # a rotated loop: its marked blocks branch back to a label BEFORE the header (the part of the body the compiler placed
    # there, unmarked) -- the body then runs from that label to the last marked block
    where = {lines[k].split(":")[0]: k for k in labels[:-1]}
    for l in lines[lo:hi]:
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and where.get(m.group(1), lo) < lo:
            lo = where[m.group(1)]
    loop = lines[lo:hi]
    ops = collections.Counter()
    for l in loop:
        m = re.match(r"([a-z_0-9]+)(\s|$)", l)
        if m and not l.startswith("."):
            ops[m.group(1)] += 1
    cls = collections.Counter()
    for op, n in ops.items():
        cls[classify(op)] += n
    total.update(cls)
    vgpr = re.search(r"; NumVgprs: (\d+)", s[j:j + 4000])
    print("role %-26s loop of %4d instructions: %s   VGPRs %s" % (ROLES[role], sum(ops.values()), dict(cls), vgpr.group(1) if vgpr else "?"))
    f64 = collections.Counter({k: v for k, v in ops.items() if "f64" in k})
    print("      f64: %s" % dict(f64.most_common()))
    oth = collections.Counter({k: v for k, v in ops.items() if k.startswith("v_") and "f64" not in k})
    print("      other VALU: %s" % dict(oth.most_common()))
    print("      LDS: %s" % {k: v for k, v in ops.items() if k.startswith("ds_")})
print("all four roles, per step: %s   VALU %d" % (dict(total), total["valu_f64"] + total["valu_other"]))
