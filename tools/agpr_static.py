"""Static look at the AGPR traffic of one kernel in a built library (diagnostic).

Disassembles the gfx950 code objects of a shared library, takes the kernel whose demangled name contains the
given text, builds its control-flow graph from the branch targets and runs a must-be-written analysis over
the AGPRs: a read of an AGPR that SOME path from the kernel entry reaches without a write is reported.
(Whole registers only -- a write under a partial EXEC mask counts as a write -- so this finds the
path-level cases, not the lane-level ones.)  tools/agpr_probe.py shows the run-time side: results that
follow the pattern a scrub kernel left in the AGPRs.

    python3 tools/agpr_static.py gogp_amd/libgogp_testhooks.so 'gogp_old::grad_reduce_kernel<64, true, double>'
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj_audit  # noqa: E402


def disassemble(so_path, want):
    for img in codeobj_audit.code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([os.path.join(codeobj_audit.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn",
                                  f.name], capture_output=True, text=True, check=True).stdout
        cur, body = None, {}
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
            if m:
                cur = m.group(1)
                body[cur] = []
            elif cur is not None and line.strip():
                body[cur].append(line.strip())
        syms = [s for s in body if not s.startswith("L") and not s.startswith(".")]
        for s, d in zip(syms, codeobj_audit.demangle(syms)):
            if want in d:
                # labels inside the kernel are separate symbols "L<n>" following it in the listing
                return d, txt, s
    return None, None, None


def kernel_lines(txt, sym):
    """[(address, text, branch target address or None)] of one kernel, in program order."""
    out, on, base = [], False, None
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:$", line)
        if m:
            if m.group(2) == sym:
                on, base = True, int(m.group(1), 16)
                continue
            if on:
                break
            continue
        if not on or not line.strip():
            continue
        m = re.match(r"^\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if not m:
            continue
        text, addr, rest = m.group(1).strip(), int(m.group(2), 16), m.group(3)
        tgt = None
        if text.startswith(("s_branch", "s_cbranch")):
            t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", rest)
            tgt = base + int(t.group(1), 16) if t else base
        out.append((addr, text, tgt))
    return out


def regs(tok):
    """a5 -> [5]; a[4:7] -> [4,5,6,7]"""
    m = re.match(r"^a(\d+)$", tok)
    if m:
        return [int(m.group(1))]
    m = re.match(r"^a\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


STORES = ("scratch_store", "global_store", "ds_write", "buffer_store", "flat_store")


def defs_uses(text):
    ops = re.split(r"[ ,]+", text)
    mnem, args = ops[0], [a for a in ops[1:] if a]
    if not any(regs(a) for a in args):
        return [], []
    dst = regs(args[0]) if not mnem.startswith(STORES) else []
    srcs = []
    for a in (args[1:] if dst else args):
        srcs += regs(a)
    return dst, srcs


def main():
    so, want = sys.argv[1], sys.argv[2]
    name, txt, sym = disassemble(so, want)
    if name is None:
        sys.exit("no kernel matching %r in %s" % (want, so))
    ins = kernel_lines(txt, sym)
    index_of = {a: i for i, (a, _, _) in enumerate(ins)}
    # basic blocks: leaders are the entry, every branch target and every instruction behind a branch / s_endpgm
    leaders = {0}
    for i, (a, t, tgt) in enumerate(ins):
        if tgt is not None:
            leaders.add(index_of[tgt])
            if i + 1 < len(ins):
                leaders.add(i + 1)
        elif t.startswith("s_endpgm") and i + 1 < len(ins):
            leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    blocks = []
    for b, st in enumerate(starts):
        en = starts[b + 1] if b + 1 < len(starts) else len(ins)
        blocks.append((st, en))
        block_of[st] = b
    succ = [[] for _ in blocks]
    for b, (st, en) in enumerate(blocks):
        a, t, tgt = ins[en - 1]
        if t.startswith("s_endpgm"):
            continue
        if tgt is not None:
            succ[b].append(block_of[index_of[tgt]])
            if t.startswith("s_branch"):
                continue
        if en < len(ins):
            succ[b].append(block_of[en])
    pred = [[] for _ in blocks]
    for b, ss in enumerate(succ):
        for x in ss:
            pred[x].append(b)
    # must-be-written analysis (bit r set: AGPR r has been written on EVERY path from the entry)
    gen = []
    for st, en in blocks:
        g = 0
        for i in range(st, en):
            for r in defs_uses(ins[i][1])[0]:
                g |= 1 << r
        gen.append(g)
    FULL = (1 << 256) - 1
    inn = [FULL] * len(blocks)
    inn[0] = 0
    out = [inn[b] | gen[b] for b in range(len(blocks))]
    changed = True
    while changed:
        changed = False
        for b in range(1, len(blocks)):
            v = FULL
            for q in pred[b]:
                v &= out[q]
            if not pred[b]:
                v = FULL  # unreachable
            if v != inn[b]:
                inn[b] = v
                o = v | gen[b]
                if o != out[b]:
                    out[b] = o
                changed = True
    nr = nw = 0
    bad = {}
    for b, (st, en) in enumerate(blocks):
        have = inn[b]
        for i in range(st, en):
            d, u = defs_uses(ins[i][1])
            for r in u:
                nr += 1
                if not (have >> r) & 1:
                    bad.setdefault(r, []).append(i)
            for r in d:
                nw += 1
                have |= 1 << r
    back = sum(1 for b, ss in enumerate(succ) for x in ss if x <= b)
    print("%s\n%d instructions, %d basic blocks, %d backward branches, %d AGPR reads, %d AGPR writes" %
          (name, len(ins), len(blocks), back, nr, nw))
    print("%d AGPRs are read at a point that some path from the kernel entry reaches without having written them "
          "(%d such reads)" % (len(bad), sum(len(v) for v in bad.values())))
    for r in sorted(bad):
        i = bad[r][0]
        print("  a%-3d %3d reads, first at +0x%x: %s" % (r, len(bad[r]), ins[i][0] - ins[0][0], ins[i][1]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
