"""Checks on gfx950 assembly (hipcc -S) of the kernels that hide their operand traffic from the compiler.

The software-pipelined kernels issue LDS-DMA (`global_load_lds_dwordx4`) and fragment reads (`ds_read_b128`, `ds_read_b64_tr_b16`)
from inline assembly and retire them with hand-COUNTED `s_waitcnt vmcnt(N)` / `lgkmcnt(N)`.  hipcc does not know those loads exist,
so three properties hold only if the generated code happens to have them -- this scan asserts them:

  resources   the kernel has no scratch (`.private_segment_fixed_size` 0) and spills no vector register: a scratch load/store inside the K loop
              is a VMEM operation the counted vmcnt would absorb in place of a DMA piece that has not landed;
  vmem        no VMEM instruction other than global_load_lds inside an innermost loop that holds both LDS-DMA and MFMAs
              (same reason, for any compiler-generated global access);
  lds         no instruction reads or writes the destination registers of a ds_read before an `s_waitcnt lgkmcnt(N)` that covers it
              (LDS returns arrive in issue order; a vector instruction is NOT interlocked against a pending LDS return), loop-
              carried cases included: a read issued at the end of one iteration against a use at the head of the next
              is seen.  The walk follows every path of the control-flow graph without evaluating branch conditions: code whose
              safety rests on two correlated branches is reported -- put an s_waitcnt in front of the re-use instead.

    python tools/exp/asm_checks.py file.s [...] [--kernels substr,substr]      # exit code 1 if anything was found
"""
import re
import sys

VREG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1):
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def kernels_of(path):
    """{kernel: [(line, text, loop)]} with labels as 'name:' entries, and {kernel: metadata}.  `loop` = (header label, is the
    header of an INNER loop) from LLVM's block comments ("in Loop: Header=BB1_17 Depth=1", "=>This Inner Loop Header: Depth=1")."""
    body, meta, cur, cur_meta = {}, {}, None, None
    loop, inner = None, set()
    last_label = None
    in_asm, manual = False, set()         # functions that issue ds_read from inline assembly (between ;;#ASMSTART and ;;#ASMEND)
    for ln, raw in enumerate(open(path), start=1):
        if ";;#ASMSTART" in raw:
            in_asm = True
        elif ";;#ASMEND" in raw:
            in_asm = False
        code, _, comment = raw.partition(";")
        t = code.strip()
        if in_asm and cur is not None and t.startswith(("ds_read", "ds_load")):
            manual.add(cur)
        m = re.match(r"    \.name:\s+(\S+)", raw)
        if m:
            cur_meta = meta.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s*\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count|group_segment_fixed_size):\s+(\d+)", raw)
        if m and cur_meta is not None:
            cur_meta[m.group(1)] = int(m.group(2))
            continue
        if cur is not None and comment and (t.endswith(":") or (not t and last_label)):
            # block comment of the label just seen (may continue on comment-only lines)
            lab = t[:-1] if t.endswith(":") else last_label
            m = re.search(r"in Loop: Header=(BB\w+)", comment)
            if m:
                loop = ".L" + m.group(1)
            if "Loop Header" in comment:
                loop = lab
                if "Inner Loop Header" in comment:
                    inner.add((cur, lab))
        if not t:
            continue
        m = re.match(r"^(_Z\w+):$", t)
        if m:
            cur, loop, last_label = m.group(1), None, None
            body[cur] = []
            continue
        if cur is None:
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
            continue
        if t.endswith(":"):
            last_label = t[:-1]
            if "Loop" not in comment:
                loop = None          # a block outside every loop (LLVM prints no loop comment for it)
            body[cur].append((ln, t, loop))
        elif not t.startswith("."):
            last_label = None
            body[cur].append((ln, t, loop))
    for k in body:
        body[k] = [(ln, t, (lp, (k, lp) in inner) if lp else None) for ln, t, lp in body[k]]
    return body, meta, manual


def is_vmem(op):
    return op.startswith(("global_", "buffer_", "scratch_", "flat_")) and not op.startswith("global_load_lds")


def check_vmem(ins):
    """innermost loops (LLVM's annotation) that hold LDS-DMA and MFMAs must hold no other VMEM instruction"""
    groups = {}
    for ln, t, lp in ins:
        if lp and lp[1] and not t.endswith(":"):
            groups.setdefault(lp[0], []).append((ln, t))
    found = []
    for head, seq in groups.items():
        ops = [t.split()[0] for _, t in seq]
        if any(o.startswith("global_load_lds") for o in ops) and any(o.startswith(("v_mfma", "v_smfma")) for o in ops):
            found += [(ln, t, f"VMEM inside the LDS-DMA + MFMA loop {head}") for ln, t in seq if is_vmem(t.split()[0])]
    return found


def blocks_of(ins):
    """basic blocks: list of (label or None, [(line, text)], [successor block indices])"""
    blocks, cur = [], [None, []]
    for ln, t, _ in ins:
        if t.endswith(":"):
            if cur[1] or cur[0] is not None:
                blocks.append(cur)
            cur = [t[:-1], []]
            continue
        cur[1].append((ln, t))
        op = t.split()[0]
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            blocks.append(cur)
            cur = [None, []]
    if cur[1] or cur[0] is not None:
        blocks.append(cur)
    index = {b[0]: i for i, b in enumerate(blocks) if b[0] is not None}
    out = []
    for i, (lab, seq) in enumerate(blocks):
        succ = []
        last = seq[-1][1].split() if seq else ["fall"]
        op = last[0]
        if op == "s_branch":
            succ = [index[last[-1]]] if last[-1] in index else []
        elif op.startswith("s_cbranch"):
            succ = ([index[last[-1]]] if last[-1] in index else []) + ([i + 1] if i + 1 < len(blocks) else [])
        elif op in ("s_endpgm",) or op.startswith(("s_setpc", "s_swappc")):
            succ = []
        elif i + 1 < len(blocks):
            succ = [i + 1]
        out.append((lab, seq, succ))
    return out


MAX_STATES = 400000


def step_lds(seq, pending, found):
    """pending: tuple of (kind, registers, line) in issue order; kind 'r' ds_read, 'w' other LDS operation"""
    pending = list(pending)
    for ln, t in seq:
        op = t.split()[0]
        if op == "s_waitcnt":
            m = LGKM.search(t)
            if m:
                n = int(m.group(1))
                if n == 0:
                    pending = []
                else:
                    # LDS operations complete in issue order; scalar memory loads share the counter but only make a counted wait
                    # stricter (outstanding <= N with more operations issued), so they are not part of the model
                    drop = len(pending) - n
                    if drop > 0:
                        del pending[:drop]
            continue
        if op.startswith("s_waitcnt"):
            continue
        touched = vregs(t[len(op):])
        if touched:
            for kind, regs, rl in pending:
                if kind == "r" and regs & touched:
                    found.add((ln, t, f"touches the destination of the ds_read at line {rl} before a covering s_waitcnt lgkmcnt"))
                    break
        if op.startswith(("ds_read", "ds_load")):
            pending.append(("r", frozenset(vregs(t[len(op):].split(",")[0])), ln))
        elif op.startswith("ds_"):
            pending.append(("w", frozenset(), ln))
        if len(pending) > 48:            # far more than the counter can express: keep the tail
            pending = pending[-48:]
    return tuple(pending)


def check_lds(ins):
    """every path through the kernel's control-flow graph (memoised on (block, outstanding LDS operations))"""
    blocks = blocks_of(ins)
    found, seen, work = set(), set(), [(0, ())]
    while work:
        if len(seen) > MAX_STATES:
            found.add((0, "", "exploration limit reached: the LDS check is incomplete for this kernel"))
            break
        b, state = work.pop()
        key = (b, tuple((k, ln) for k, _, ln in state))
        if key in seen:
            continue
        seen.add(key)
        _, seq, succ = blocks[b]
        out = step_lds(seq, state, found)
        for s2 in succ:
            work.append((s2, out))
    return sorted(found)


# kernels that retire LDS-DMA with hand-counted vmcnt: a spill or any scratch traffic would be absorbed by the count
COUNTED_VMCNT = ("igemm_pipe_kernel", "wgemm_pipe_kernel", "brig_kernel", "head_softmax_kernel", "head_dgrad_kernel")


def check_file(path, want=None):
    """-> (findings, notes, names).  findings: resource use of the counted-vmcnt kernels, VMEM inside an LDS-DMA + MFMA loop,
    a touch of a pending ds_read destination INSIDE a loop.  notes: the same touch outside every loop -- behind a K loop the walk
    cannot tell that "the last iteration issues no read" and "the loop ends" are one condition; the kernels carry an explicit
    s_waitcnt there, but hipcc is free to place accumulator copies in front of it."""
    body, meta, manual = kernels_of(path)
    bad, notes = [], []
    for name, ins in body.items():
        if want and not any(w in name for w in want):
            continue
        md = meta.get(name)
        if md is not None and any(k in name for k in COUNTED_VMCNT):
            for key in ("private_segment_fixed_size", "vgpr_spill_count"):       # (SGPR spills go to VGPR lanes: no memory traffic)
                if md.get(key, 0) != 0:
                    bad.append((name, 0, "", f"{key} = {md[key]}"))
        in_loop = {ln for ln, t, lp in ins if lp is not None}
        seen_lines = set()
        # the LDS walk is for the functions whose fragment reads the compiler cannot see (it waits correctly for its own)
        for ln, t, why in check_vmem(ins) + (check_lds(ins) if name in manual else []):
            kind = why.split(" at line")[0]
            if (ln, kind) in seen_lines:         # one report per instruction and kind
                continue
            seen_lines.add((ln, kind))
            (bad if (ln in in_loop or ln == 0 or kind.startswith("VMEM")) else notes).append((name, ln, t, why))
    return bad, notes, [n for n in body if not want or any(w in n for w in want)]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want = None
    for i, a in enumerate(sys.argv):
        if a == "--kernels":
            want = sys.argv[i + 1].split(",")
            args = [x for x in args if x != sys.argv[i + 1]]
    total = 0
    for f in args:
        bad, notes, names = check_file(f, want)
        print(f"{f.split('/')[-1]}: {len(names)} functions checked, {len(bad)} findings, {len(notes)} unproven touches outside loops")
        for name, ln, t, why in bad:
            print(f"  {name}: line {ln}: {t}   <- {why}")
        if "--notes" in sys.argv:
            for name, ln, t, why in notes:
                print(f"  (note) {name}: line {ln}: {t}   <- {why}")
        total += len(bad)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
