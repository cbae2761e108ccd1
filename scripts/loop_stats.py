#!/usr/bin/env python3
"""Instruction mix of the inner sweep loop(s) of one solve instantiation (development aid)."""
import collections, re, subprocess, sys, os
nr, nc = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("1", "2")
nb0 = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3].isdigit() else "0"   # last complex term known to have b = 0
extra = [a for a in sys.argv[3:] if not a.isdigit()]
os.makedirs("/tmp/isa", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-c",
                "/root/repo/mind_the_gaps_amd/csrc/mtg_kernels.hip", "-save-temps", "-o", "/tmp/isa/k.o"],
               cwd="/tmp/isa", stderr=subprocess.DEVNULL)
s = open("/tmp/isa/mtg_kernels-hip-amdgcn-amd-amdhsa-gfx950.s").read()
name = "_Z16mtg_solve_kernelILi%sELi%sELi%sEEv12MtgSolveArgs" % (nr, nc, nb0)
i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
lines = [l.strip() for l in s[i:j].split("\n")]
heads = [k for k, l in enumerate(lines) if "Loop Header" in l]
for h in heads:
    lab = lines[h - 1].split(":")[0] if lines[h - 1].startswith(".LBB") else lines[h].split(":")[0]
    ends = [k for k, l in enumerate(lines) if k > h and re.search(r"branch\w*\s+" + re.escape(lab) + r"\b", l)]
    if not ends:
        continue
    loop = lines[h:ends[-1] + 1]
    cnt = collections.Counter()
    for l in loop:
        m = re.match(r"([a-z_0-9]+)\s", l)
        if m:
            cnt[m.group(1)] += 1
    valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
    f64 = sum(v for k, v in cnt.items() if "f64" in k)
    print(lab, "instrs", sum(cnt.values()), "VALU", valu, "f64", f64)
    print("  ", dict(cnt.most_common(50)))
    open("/tmp/isa/loop_%s_%s_%d.s" % (nr, nc, h), "w").write("\n".join(loop))
