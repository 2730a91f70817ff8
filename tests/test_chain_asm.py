"""Static check of the chained layer-1 kernel's generated code (no GPU needed: hipcc cross-compiles to assembly).

l1_chain.hip issues its streaming loads as inline asm the compiler does not track and waits for them with ONE hand-counted
`s_waitcnt vmcnt(12)` per iteration.  The compiler believes an asm output is valid immediately, so nothing stops it from
copying or reusing a destination register while the load is still in flight (a phi copy after a load under a branch, a
live-range split under register pressure) - a silent, timing-dependent corruption no parity test is sure to catch.  This
test reads the generated assembly and asserts that between every untracked load and the next hand-counted wait no
compiler-generated instruction mentions the load's destination registers, that the loop is drained before the epilogue,
and that the loop really is pipelined (the waits inside it allow 12 outstanding operations)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _kernel_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="chain_asm_"), "l1_chain.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "locator_amd", "csrc", "l1_chain.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    shutil.rmtree(os.path.dirname(out), ignore_errors=True)
    m = re.search(r"^(_Z24l1_bwd_adam_chain_kernelILi13E\w*):\s.*?^\s*s_endpgm", text, re.S | re.M)
    assert m, "kernel instantiation <13> not found in the assembly"
    return m.group(0).splitlines()


def _regs(tok):
    """VGPR numbers named by an operand token: v12 or v[12:15]."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _parse(lines):
    """-> list of (kind, text, regs): kind in {'aload', 'astore', 'await', 'ins', 'label'}; asm = between ASMSTART / ASMEND."""
    out, in_asm = [], False
    for ln in lines:
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                out.append(("label", s.split(":")[0], set()))
            continue
        s = s.split(";")[0].strip()
        toks = re.split(r"[\s,]+", s)
        regs = set()
        for t in toks[1:]:
            regs |= _regs(t)
        if in_asm and toks[0].startswith("global_load"):
            out.append(("aload", s, _regs(toks[1])))
        elif in_asm and toks[0].startswith("global_store"):
            out.append(("astore", s, regs))
        elif in_asm and toks[0] == "s_waitcnt" and "vmcnt" in s:
            out.append(("await", s, set()))
        else:
            out.append(("ins", s, regs))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_instruction_touches_an_untracked_load_destination_before_the_hand_counted_wait():
    prog = _parse(_kernel_asm())
    labels = {t: i for i, (k, t, _) in enumerate(prog) if k == "label"}
    # the main loop: the backward branch that spans the most untracked loads
    best = None
    for i, (k, t, _) in enumerate(prog):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t) if k == "ins" else None
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < i:
                n = sum(1 for kk, _, _ in prog[tgt:i] if kk == "aload")
                if best is None or n > best[0]:
                    best = (n, tgt, i)
    assert best and best[0] >= 2 * (12 + 3), f"main loop with its 2 x (12 + 3) untracked loads not found: {best}"
    _, lo, hi = best
    # walk: prologue + loop body, then the loop body once more (a load at the end of the body is awaited at its top)
    seq = prog[:hi + 1] + prog[lo:hi + 1]
    pending = []                                   # untracked loads in flight, oldest first: (text, destination registers)
    n_loads = n_waits = 0
    for kind, text, regs in seq:
        if kind == "aload":
            n_loads += 1
            pending.append((text, regs))
        elif kind == "await":
            n_waits += 1
            n = int(re.search(r"vmcnt\((\d+)\)", text).group(1))
            pending = pending[-n:] if n else []    # loads retire in order: "at most n outstanding" leaves the n newest
        elif kind in ("ins", "astore"):
            for ltext, lregs in pending:
                bad = regs & lregs
                assert not bad, f"`{text}` touches v{sorted(bad)} while `{ltext}` is in flight"
    assert n_loads >= 3 * (12 + 3) and n_waits >= 4
    # the loop's waits are the pipelined ones, and the loop is drained before the epilogue reuses registers
    in_loop = [t for k, t, _ in prog[lo:hi + 1] if k == "await"]
    assert in_loop and all("vmcnt(12)" in t for t in in_loop), in_loop
    after = [t for k, t, _ in prog[hi + 1:] if k == "await"]
    assert any("vmcnt(0)" in t for t in after), "no drain of the untracked loads after the loop"
