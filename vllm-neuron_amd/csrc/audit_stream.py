#!/usr/bin/env python3
"""Audit the ISA hipcc emits for the hand-counted weight stream (mi_common.h stream_load16 /
stream_wait).

Between an inline-asm `global_load_dwordx4 v[a:b] ... ; stream_load` and the
`s_waitcnt ... ; stream_release v[a:b]` that retires it, NO instruction may touch v[a:b]: a
compiler copy, spill or reuse there would move bytes that have not landed yet (wrong weights,
or a memory fault if the register is reused as an address).  Kernels that use the stream must
not use scratch either.

The check is a may-analysis over the kernel's control-flow graph: a register range is
"in flight" at a point if it is in flight on ANY path reaching it.  build.py runs this on
every build and fails the build on a finding.

    python audit_stream.py linear_kernels.s
"""
import re
import sys

VREG = re.compile(r"v\[\d+:\d+\]|v\d+")


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return frozenset({int(m.group(1))}) if m else frozenset()


def split_kernels(path):
    name, body = None, []
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\S*gemv_kernel\S*):", t)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if t.startswith(".Lfunc_end"):
            yield name, body
            name = None
            continue
        body.append((ln, t))


def blocks_of(body):
    """-> list of (label, [(ln, text)]), dict label -> index"""
    blocks, cur, label = [], [], "<entry>"
    for ln, t in body:
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        if not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
            continue
        cur.append((ln, t))
        if re.match(r"^(s_branch|s_cbranch_\w+)\s", t):   # a branch ends its basic block
            blocks.append((label, cur))
            label, cur = f"<anon{ln}>", []
    blocks.append((label, cur))
    return blocks, {lab: i for i, (lab, _) in enumerate(blocks)}


def transfer(block, pending, report):
    pending = dict(pending)
    for ln, t in block:
        code = t.split(";")[0]
        toks = VREG.findall(code)
        if "stream_load" in t:
            dst = regs(toks[0])
            addr = frozenset().union(*[regs(x) for x in toks[1:]]) if len(toks) > 1 else frozenset()
            for p, pl in pending.items():
                if p & (dst | addr):
                    report(ln, f"stream_load touches registers still in flight from line {pl}: {t}")
            pending[dst] = ln
        elif "stream_release" in t:
            rel = regs(VREG.findall(t.split("stream_release")[1])[0])
            pending.pop(rel, None)
        else:
            touched = frozenset().union(*[regs(x) for x in toks]) if toks else frozenset()
            for p, pl in pending.items():
                if p & touched:
                    report(ln, f"touches v{min(p)}..v{max(p)} in flight since line {pl}: {t}")
            if "scratch_" in code:
                report(ln, f"scratch access in a stream kernel: {t}")
    return pending


def audit_kernel(name, body):
    blocks, index = blocks_of(body)
    succ = []
    for i, (_, blk) in enumerate(blocks):
        s, falls = [], True
        for _, t in blk:
            code = t.split(";")[0]
            m = re.match(r"^(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", code)
            if m:
                s.append(index[m.group(2)])
                if m.group(1) == "s_branch":
                    falls = False
        if blk and blk[-1][1].startswith("s_endpgm"):
            falls = False
        if falls and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    entry = [dict() for _ in blocks]
    work, seen_once = [0], set()
    problems = {}
    while work:
        i = work.pop()
        out = transfer(blocks[i][1], entry[i], lambda ln, msg: problems.setdefault((ln, msg), None))
        for j in succ[i]:
            merged = dict(entry[j])
            merged.update({k: v for k, v in out.items() if k not in merged})
            if merged.keys() != entry[j].keys() or j not in seen_once:
                entry[j] = merged
                seen_once.add(j)
                work.append(j)
    return [f"{name[:70]}:{ln}: {msg}" for (ln, msg) in sorted(problems)]


def audit(path):
    n, problems = 0, []
    for name, body in split_kernels(path):
        n += 1
        if any("stream_load" in t for _, t in body):
            problems += audit_kernel(name, body)
    return n, problems


if __name__ == "__main__":
    k, probs = audit(sys.argv[1])
    for p in probs[:40]:
        print(p)
    print(f"audited {k} gemv kernels: {len(probs)} problems")
    sys.exit(1 if probs or k == 0 else 0)
