"""not gpu: properties of the COMPILED kernels that the source cannot show (hipcc -S of every source file that holds MFMAs, ~1 min).

1. tools/exp/store_mfma_hazard.py: no global store whose data registers a following MFMA writes back before the store is known to
   have read them (measured in round 4: hipcc pads that pair for vector instructions only, and the matrix pipe's write-back is
   not interlocked against the store's read).
2. tools/exp/asm_checks.py: the software-pipelined kernels hide their operand traffic from hipcc (LDS-DMA and fragment
   reads from inline assembly, hand-counted s_waitcnt).  Asserted on the generated code: no scratch and no vector spills in those
   kernels, no other VMEM instruction inside a loop that holds LDS-DMA and MFMAs, no instruction inside a loop that touches the
   destination of an inline-assembly ds_read before a covering s_waitcnt lgkmcnt (every path of the control-flow graph)."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "palette_and_histo_gan_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
sys.path.insert(0, os.path.join(ROOT, "tools", "exp"))


def mfma_sources():
    """every csrc/*.hip that issues MFMAs (derived, not a hand list: a new kernel file is scanned without anybody remembering to
    add it -- VERDICT r04 weak #10), longest compile first"""
    names = [f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip") and re.search(r"mfma", open(os.path.join(CSRC, f)).read())]
    return sorted(names, key=lambda n: -os.path.getsize(os.path.join(CSRC, n + ".hip")))


def _asm(name, out_dir):
    out = os.path.join(out_dir, name + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, os.path.join(CSRC, name + ".hip"), "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


@pytest.fixture(scope="module")
def asm_files(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("asm"))
    names = mfma_sources()
    assert {"igemm", "brig", "wgemm", "hist", "wgrad_small", "conv_strip", "conv_fewin", "conv_fewout", "head_softmax"} <= set(names), names
    with ThreadPoolExecutor(max_workers=4) as pool:
        return list(pool.map(lambda n: _asm(n, out), names))


def test_no_mfma_writes_back_into_the_data_registers_of_a_pending_store(asm_files):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp", "store_mfma_hazard.py")] + asm_files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert all(f"{os.path.basename(f)}: 0 store" in r.stdout for f in asm_files), r.stdout


def test_hand_counted_waits_match_the_generated_code(asm_files):
    import asm_checks as A
    manual_total = loops_total = 0
    for f in asm_files:
        bad, notes, names = A.check_file(f)
        assert not bad, "\n".join(f"{os.path.basename(f)}: {n}: line {ln}: {t} <- {why}" for n, ln, t, why in bad)
        body, _, manual = A.kernels_of(f)
        manual_total += len(manual)
        for ins in body.values():
            groups = {}
            for ln, t, lp in ins:
                if lp and lp[1] and not t.endswith(":"):
                    groups.setdefault(lp[0], []).append(t.split()[0])
            loops_total += sum(1 for g in groups.values() if any(o.startswith("global_load_lds") for o in g) and any(o.startswith("v_mfma") for o in g))
    # the scan is not vacuous: the pipelined kernels are there and were recognised
    assert manual_total >= 20 and loops_total >= 50, (manual_total, loops_total)


BAD_ASM = """
\t.text
_Z3badv:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\t;;#ASMSTART
\tglobal_load_lds_dwordx4 v1, s[0:1]
\t;;#ASMEND
\t;;#ASMSTART
\tds_read_b128 v[2:5], v0 offset:0
\t;;#ASMEND
\t;;#ASMSTART
\tds_read_b128 v[6:9], v0 offset:64
\t;;#ASMEND
\ts_waitcnt lgkmcnt(1)
\tv_mfma_f32_32x32x16_bf16 a[0:15], v[2:5], v[2:5], a[0:15]
\tv_add_u32_e32 v10, v6, v0
\tglobal_load_dword v11, v[12:13], off
\ts_waitcnt lgkmcnt(0)
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
    .name:           _Z3badv
    .private_segment_fixed_size: 0
    .vgpr_spill_count: 0
"""


def test_the_scanner_sees_what_it_is_there_for(tmp_path):
    """a loop that (a) uses the second fragment while only the first is covered by the counted wait and (b) holds a plain global load
    beside its LDS-DMA: both must be reported, and the covered use of the first fragment must not"""
    import asm_checks as A
    p = tmp_path / "bad.s"
    p.write_text(BAD_ASM)
    bad, notes, names = A.check_file(str(p))
    whys = [(t.split()[0], why) for _, _, t, why in bad]
    assert ("v_add_u32_e32", "touches the destination of the ds_read at line 13 before a covering s_waitcnt lgkmcnt") in whys, whys
    assert any(op == "global_load_dword" and why.startswith("VMEM inside") for op, why in whys), whys
    assert not any(op.startswith("v_mfma") for op, _ in whys), whys
