"""Scan gfx950 assembly (hipcc -S) for a global/buffer store whose DATA (or address) registers a later v_mfma writes before any
`s_waitcnt vmcnt(0)`.

Why: measured in round 4 (tools/exp/hist_repro.py, rgbuv_hist_bwd3_kernel): `global_store_dwordx4 v[0:1], v[2:5]` followed, a dozen
instructions and an LDS wait later, by `v_mfma_f32_32x32x16_bf16 v[0:15], ...` stored a wrong first dword for the wave's last 16 lanes
about once per 1 000 workgroups -- the store had not read its data registers when the matrix pipe wrote them back.  hipcc pads
this pair for vector instructions only.  A `s_waitcnt vmcnt(0)` after the store removed it (0 of 399 launches against 74 of 399).

    python tools/exp/store_mfma_hazard.py file.s [...]          # prints each (kernel, store line, mfma line, distance)

The scan follows the text order, falls through conditional branches, and follows each backward branch once (a store at the end of
a loop body against an MFMA at its head)."""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
WINDOW = 1500          # instructions
LDS_TOO = False        # --lds: the same question for ds_write data registers (until lgkmcnt(0))


def regs(tok):
    m = REG.search(tok)
    if not m:
        return None
    if m.group(1) is not None:
        return (int(m.group(1)), int(m.group(2)))
    return (int(m.group(3)), int(m.group(3)))


def overlap(a, b):
    return a is not None and b is not None and a[0] <= b[1] and b[0] <= a[1]


def scan_kernel(name, lines):
    labels = {}
    ins = []
    for ln, t in lines:
        t = t.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            labels[t[:-1]] = len(ins)
            continue
        if t.startswith("."):
            continue
        ins.append((ln, t))
    hits = []
    for i, (ln, t) in enumerate(ins):
        op = t.split()[0]
        is_lds = op.startswith("ds_write")
        if not (is_lds or op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store") or op.startswith("scratch_store")):
            continue
        if is_lds and not LDS_TOO:
            continue
        ops = [o.strip() for o in t[len(op):].split(",")]
        if op.startswith("buffer_store"):
            data, addr = regs(ops[0]), regs(ops[1]) if len(ops) > 1 else None
        else:
            addr, data = regs(ops[0]), regs(ops[1]) if len(ops) > 1 else None
            if is_lds and len(ops) > 2 and regs(ops[2]) and "offset" not in ops[2]:      # ds_write2: two data operands
                d2 = regs(ops[2])
                data = (min(data[0], d2[0]), max(data[1], d2[1])) if data else d2
        j, steps, jumped = i + 1, 0, set()
        # registers the store may still have to read; a vector instruction's or an LDS/VMEM return's write to one of them is
        # interlocked by the hardware (it waits for the read), so it takes the register off the list
        pending = set(range(data[0], data[1] + 1)) if data else set()
        while j < len(ins) and steps < WINDOW and pending:
            lj, tj = ins[j]
            oj = tj.split()[0]
            if oj == "s_endpgm":
                break
            if oj == "s_waitcnt" and ("lgkmcnt(0)" if is_lds else "vmcnt(0)") in tj:
                break
            if is_lds and oj == "s_barrier":
                break
            if oj.startswith("v_mfma") or oj.startswith("v_smfma"):
                dst = regs(tj[len(oj):].split(",")[0])
                if dst and pending & set(range(dst[0], dst[1] + 1)):
                    hits.append((name, ln, t, lj, tj, steps, "data"))
                    break
            elif oj.startswith("v_") or oj.startswith("ds_read") or oj.startswith("global_load") or oj.startswith("buffer_load"):
                first = tj[len(oj):].split(",")[0]
                dst = regs(first)
                if dst and not (oj.startswith("global_load") and "lds" in oj):
                    pending -= set(range(dst[0], dst[1] + 1))
            if oj in ("s_branch",) or oj.startswith("s_cbranch"):
                tgt = tj.split()[-1]
                if tgt in labels and labels[tgt] <= j and tgt not in jumped:
                    jumped.add(tgt)
                    if oj == "s_branch":
                        j = labels[tgt]
                        continue
                    # conditional backward branch: follow it (the loop case); the fall-through is scanned from later stores anyway
                    j = labels[tgt]
                    continue
                if oj == "s_branch" and tgt in labels:
                    j = labels[tgt]
                    continue
            j += 1
            steps += 1
    return hits


def main():
    global LDS_TOO
    total = 0
    if "--lds" in sys.argv:
        LDS_TOO = True
        sys.argv.remove("--lds")
    for path in sys.argv[1:]:
        cur, buf, out = None, [], []
        for ln, t in enumerate(open(path), 1):
            m = re.match(r"^(_Z\w+):", t)
            if m:
                cur, buf = m.group(1), []
                continue
            if cur:
                buf.append((ln, t))
                if "s_endpgm" in t and ".Lfunc_end" not in t:
                    pass
                if t.startswith(".Lfunc_end"):
                    out += scan_kernel(cur, buf)
                    cur = None
        print(f"{path}: {len(out)} store -> MFMA register reuse(s) without vmcnt(0) in between")
        seen = set()
        for name, ln, t, lj, tj, steps, what in out:
            key = (name, ln)
            if key in seen:
                continue
            seen.add(key)
            print(f"  {name[:60]}  line {ln}: {t}   ->  line {lj}: {tj.split(',')[0]}  ({steps} instructions later, {what})")
        total += len(out)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
