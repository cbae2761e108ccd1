"""Lint the gfx950 assembly of the library's kernels for accumulator registers (AGPRs) that are read
although NO path from the kernel's entry writes them.

Why: under register pressure the allocator parks values in AGPRs, and a kernel that also spills to
scratch has been seen (rocm 7.2 clang, the J = 6 time-parallel kernel) to reload the low half of a
64-bit loop bound from scratch into one AGPR and to read the high half from the neighbouring AGPR,
which nothing had written: the bound was made of whatever the previous wave left there, the loop ran
off the end of the light curve and the GPU faulted -- sometimes.  A forward may-be-written dataflow
over the control-flow graph finds exactly such reads; `make lint` runs it over every translation unit.

Usage: agpr_lint.py file.s [file.s ...]     (exit status 1 when anything is found)"""
import re
import sys

WRITERS = ("v_accvgpr_write", "v_accvgpr_mov", "scratch_load", "ds_read", "global_load", "buffer_load", "v_mfma")


def regs(tok):
    m = re.match(r"a\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"a(\d+)$", tok)
    return [int(m.group(1))] if m else []


def functions(path):
    fn, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if fn and body:
                yield fn, body
            fn, body = m.group(1), []
            continue
        if line.startswith(".Lfunc_end"):
            if fn and body:
                yield fn, body
            fn, body = None, []
            continue
        if fn is not None:
            body.append(line)
    if fn and body:
        yield fn, body


def lint(fn, body):
    # basic blocks
    blocks, labels, cur = [], {}, []
    for line in body:
        text = line.split(";")[0].strip()
        if not text:
            continue
        m = re.match(r"^(\.LBB\w+):", text)
        if m:
            if cur:
                blocks.append(cur)
            cur = []
            labels[m.group(1)] = len(blocks)
            continue
        if text.startswith("."):
            continue
        cur.append(text)
        if text.startswith(("s_branch", "s_cbranch", "s_endpgm")):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1] if b else ""
        s = []
        if last.startswith("s_endpgm"):
            pass
        elif last.startswith("s_branch"):
            s.append(labels.get(last.split()[-1]))
        elif last.startswith("s_cbranch"):
            s.append(labels.get(last.split()[-1]))
            s.append(i + 1)
        else:
            s.append(i + 1)
        succ.append([x for x in s if x is not None and x < len(blocks)])
    # per block: registers read before being written in the block, registers written
    use, gen = [], []
    for b in blocks:
        u, g = [], set()
        for ins in b:
            op, _, rest = ins.partition(" ")
            ops = [o.strip() for o in rest.split(",")]
            writes = op.startswith(WRITERS)
            for tok in (ops[1:] if writes or op.startswith("v_accvgpr_read") else ops):
                for r in regs(tok):
                    if r not in g:
                        u.append((r, ins))
            if writes and ops:
                g.update(regs(ops[0]))
        use.append(u)
        gen.append(g)
    # may-be-written at block entry: union over predecessors, to a fixed point
    n = len(blocks)
    inn = [set() for _ in range(n)]
    work = [0] if n else []
    seen = {0}
    while work:
        i = work.pop()
        out = inn[i] | gen[i]
        for j in succ[i]:
            if j not in seen or not out <= inn[j]:
                seen.add(j)
                inn[j] |= out
                work.append(j)
    found = []
    for i in range(n):
        if i not in seen:
            continue
        for r, ins in use[i]:
            if r not in inn[i]:
                found.append("%s: a%d is read but never written on any path: %s" % (fn, r, ins))
    return found


total = 0
for path in sys.argv[1:]:
    for fn, body in functions(path):
        for msg in lint(fn, body):
            print("%s: %s" % (path, msg))
            total += 1
print("%d accumulator registers read without a write" % total)
sys.exit(1 if total else 0)
