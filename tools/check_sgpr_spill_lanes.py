#!/usr/bin/env python3
"""Static check of AMDGPU SGPR-spill lanes in one kernel of a `--save-temps` .s file.

LLVM spills SGPRs into lanes of dedicated VGPRs (`v_writelane_b32` / `v_readlane_b32`); under register pressure those VGPRs are
themselves copied or spilled to scratch in whole-wave mode.  This script runs a forward must-be-defined analysis over the kernel's
control-flow graph for every (VGPR, lane) pair and every (scratch slot, lane) pair and reports each `v_readlane_b32` whose lane is
not written on every path that reaches it -- i.e. an SGPR restored from a lane that may hold garbage.  It also reports vector
instructions that touch a spill-lane VGPR outside a `s_or_saveexec_b64 ..., -1` region.

    python tools/check_sgpr_spill_lanes.py file.s mangled_kernel_name
"""
import re
import sys


def kernel_lines(path, name):
    out, on = [], False
    for l in open(path):
        if not on and l.startswith(name) and l.rstrip().endswith(":") or (not on and l.startswith(name + ":")):
            on = True
        if on:
            if l.startswith(".Lfunc_end"):           # (not the first s_endpgm: a kernel may end in several places)
                break
            out.append(l.rstrip("\n"))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    L = kernel_lines(path, name)
    assert L, "kernel not found"
    # ---- basic blocks
    labels, blocks, cur = {}, [], None
    for i, l in enumerate(L):
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m or cur is None:
            cur = {"start": i, "ins": [], "succ": [], "label": m.group(1) if m else None}
            if m:
                labels[m.group(1)] = len(blocks)
            blocks.append(cur)
            if m:
                continue
        if s.startswith(";") and "implicit-def: $vgpr" in s:
            cur["ins"].append((i, s))
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur["ins"].append((i, s.split(";")[0].strip()))
        if s.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            cur = None if False else {"start": i + 1, "ins": [], "succ": [], "label": None}
            blocks.append(cur)
    for bi, b in enumerate(blocks):
        last = b["ins"][-1][1] if b["ins"] else ""
        if last.startswith("s_endpgm"):
            continue
        if last.startswith("s_branch"):
            b["succ"] = [labels[last.split()[1]]]
            continue
        if last.startswith("s_cbranch"):
            b["succ"].append(labels[last.split()[1]])
        if bi + 1 < len(blocks):
            b["succ"].append(bi + 1)
    lane_regs = set()
    for b in blocks:
        for _, s in b["ins"]:
            m = re.match(r"v_writelane_b32 (v\d+),", s)
            if m:
                lane_regs.add(m.group(1))
    # ---- transfer function; state: dict loc -> frozenset of defined lanes (missing = none defined); None = unreached (top)
    findings = []

    def transfer(b, st, report):
        st = dict(st)
        wwm = False
        for i, s in b["ins"]:
            m = re.search(r"implicit-def: \$vgpr(\d+)", s)
            if m:
                st["v" + m.group(1)] = frozenset()
                continue
            if re.match(r"s_or_saveexec_b64 s\[\d+:\d+\], -1", s):
                wwm = True
                continue
            if s.startswith(("s_mov_b64 exec", "s_and_saveexec", "s_or_b64 exec", "s_andn2_b64 exec", "s_xor_b64 exec")):
                wwm = False
                continue
            m = re.match(r"v_writelane_b32 (v\d+), \S+ (\d+)", s)
            if m:
                st[m.group(1)] = st.get(m.group(1), frozenset()) | {int(m.group(2))}
                continue
            m = re.match(r"v_readlane_b32 (\S+), (v\d+), (\d+)", s)
            if m:
                if m.group(2) not in lane_regs:          # readlane of an ordinary VGPR (wave reductions)
                    continue
                if report and int(m.group(3)) not in st.get(m.group(2), frozenset()):
                    findings.append((i + 1, "readlane of a lane not written on every path", s))
                continue
            m = re.match(r"scratch_store_dword(x\d)? off, (v\d+|v\[\d+:\d+\]), off(?: offset:(\d+))?", s)
            if m and not (set(re.findall(r"\bv\d+\b", s)) & lane_regs):
                n = int((m.group(1) or "x1")[1:])
                for k in range(n):
                    st["vslot%d" % (int(m.group(3) or 0) + 4 * k)] = frozenset([0])
                continue
            m = re.match(r"scratch_load_dword(x\d)? (v\d+|v\[\d+:\d+\]), off, off(?: offset:(\d+))?", s)
            if m and not (set(re.findall(r"\bv\d+\b", s)) & lane_regs):
                n = int((m.group(1) or "x1")[1:])
                for k in range(n):
                    if report and "vslot%d" % (int(m.group(3) or 0) + 4 * k) not in st:
                        findings.append((i + 1, "reload of a VGPR spill slot not stored on every path", s))
                continue
            regs = set(re.findall(r"\bv\d+\b", s))
            for mm in re.finditer(r"v\[(\d+):(\d+)\]", s):
                regs |= {"v%d" % k for k in range(int(mm.group(1)), int(mm.group(2)) + 1)}
            if not (regs & lane_regs):
                continue
            if report and not wwm:
                findings.append((i + 1, "vector access to a spill-lane VGPR outside whole-wave mode", s))
            m = re.match(r"scratch_store_dword off, (v\d+), off(?: offset:(\d+))?", s)
            if m:
                st["slot" + (m.group(2) or "0")] = st.get(m.group(1), frozenset())
                continue
            m = re.match(r"scratch_load_dword (v\d+), off, off(?: offset:(\d+))?", s)
            if m:
                st[m.group(1)] = st.get("slot" + (m.group(2) or "0"), frozenset())
                continue
            m = re.match(r"v_mov_b32_e32 (v\d+), (v\d+)$", s)
            if m:
                st[m.group(1)] = st.get(m.group(2), frozenset())
                continue
            if report:
                findings.append((i + 1, "unmodelled instruction on a spill-lane VGPR", s))
        return st

    def meet(a, b):
        if a is None:
            return b
        if b is None:
            return a
        return {k: a[k] & b[k] for k in a.keys() & b.keys() if a[k] & b[k]}

    IN = [None] * len(blocks)
    IN[0] = {}
    work = [0]
    while work:
        bi = work.pop()
        out = transfer(blocks[bi], IN[bi], False)
        for s in blocks[bi]["succ"]:
            new = meet(IN[s], out)
            if IN[s] is None or new != IN[s]:
                IN[s] = new
                work.append(s)
    for bi, b in enumerate(blocks):
        if IN[bi] is not None:
            transfer(b, IN[bi], True)
    print("%s: %d blocks, spill-lane VGPRs %s, %d findings" % (name, len(blocks), sorted(lane_regs, key=lambda r: int(r[1:])), len(findings)))
    for f in sorted(set(findings))[:60]:
        print("  line %d: %s: %s" % f)


if __name__ == "__main__":
    main()
