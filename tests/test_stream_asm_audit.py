"""Audit of the inline-asm prologue loads of the decode stream kernel (arcquant_amd/csrc/gemm_stream.hip; the packed-activation
instantiations -- the fused ones await their prologue loads BEFORE the first weight load and use no asm load at all).

hipcc does not track an `asm volatile` load: the destination registers count as written at the end of the statement, long
before the data lands.  The kernel retires those loads itself (asm_wait_behind_units + asm_tie).  This test compiles the
translation unit to gfx950 assembly and checks, for every instantiation, along the instruction stream from each asm load to
the hand-written `s_waitcnt vmcnt(N)` that retires it, that NO instruction touches the load's destination registers (a
compiler-inserted copy or spill of an in-flight register would read garbage: cdna_hip_programming.md 5.7 item 1), that the
hand-counted wait `vmcnt(3 c)` matches what hipcc emitted between the asm loads and the wait -- six conditional units of exactly
three loads each (2 x dwordx4 + 1 x dword), no store, nothing else in the vector-memory queue -- and that the kernels
neither spill nor use scratch."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "arcquant_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_instruction_touches_an_in_flight_asm_load(tmp_path):
    asm = tmp_path / "gemm_stream.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-I", CSRC, os.path.join(CSRC, "gemm_stream.hip"), "-o", str(asm)], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN4arcq18gemm_stream_kernel\w+):", text, flags=re.M)
    assert len(kernels) >= 8
    for name in kernels:
        a = text.index("\n" + name + ":")
        body = text[a:text.index("s_endpgm", a)].split("\n")
        in_asm, pending, n_loads, n_waits = False, set(), 0, 0
        between, seen_wait = [], False                     # compiler VMEM instructions from the last asm load to the first asm wait
        for line in body:
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            if in_asm and t.startswith("global_load"):
                dst = t.split(None, 1)[1].split(",")[0]
                assert not (_regs(t.split(",", 1)[1]) & pending), f"{name}: asm load address uses an in-flight register: {t}"
                pending |= _regs(dst)
                n_loads += 1
                between = []
                continue
            if in_asm and t.startswith("s_waitcnt") and "vmcnt" in t:
                n_waits += 1
                seen_wait = True
                continue                                   # one arm of the counted wait; the registers are released by the tie below
            if in_asm:
                continue
            if pending and n_waits:                        # first compiler instruction after the hand-written wait: all retired
                pending, n_waits = set(), 0
            touched = _regs(t) & pending
            assert not touched, f"{name}: `{t}` touches in-flight asm-load registers {sorted(touched)}"
            if pending and not seen_wait and re.match(r"(global|buffer|flat|scratch)_", t):
                between.append(t.split()[0])
        packed = "ILi0E" in name                          # gemm_stream_kernel<kSrcPacked, ...>
        if not packed:
            assert n_loads == 0, f"{name}: the fused sources await their prologue loads before the weights: no asm load expected"
        else:
            assert n_loads >= 3, name
            # asm_wait_behind_units(c) waits for vmcnt(3 c): exactly three loads per conditional unit, six units, nothing else
            assert between == ["global_load_dwordx4", "global_load_dwordx4", "global_load_dword"] * 6, f"{name}: {between}"
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta), name + ": scratch in use"
