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


def _kernels(source, pattern, defines=()):
    """{mangled name: assembly lines} of the kernels of locator_amd/csrc/<source> whose name matches `pattern` (defines: extra
    -D switches, for checking a measurement build)."""
    out = os.path.join(tempfile.mkdtemp(prefix="asm_check_"), "k.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                           *[f"-D{d}" for d in defines],
                           os.path.join(ROOT, "locator_amd", "csrc", source), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    shutil.rmtree(os.path.dirname(out), ignore_errors=True)
    res = {}
    # a kernel = its label up to the end-of-function marker (NOT the first s_endpgm: a workgroup that leaves early - the surplus
    # workgroups of l1_gemm_i8_kernel's group mapping, round 5 - has one in the middle of the function)
    for m in re.finditer(r"^(_Z\w+):\s.*?^\.Lfunc_end\d+:", text, re.S | re.M):
        if re.search(pattern, m.group(1)):
            res[m.group(1)] = m.group(0).splitlines()
    return res


def _kernel_asm(nht=8, rb=1, defines=()):
    ks = _kernels("l1_chain.hip", r"^_Z24l1_bwd_adam_chain_kernelILi13ELi%dELi%dEE" % (nht, rb), defines)
    assert len(ks) == 1, f"kernel instantiation <13, {nht}, {rb}> not found in the assembly"
    return next(iter(ks.values()))


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
        if in_asm and toks[0].startswith("global_load_lds"):
            out.append(("aload", s, set()))        # LDS-DMA: counted in vmcnt, no register destination (operand = address)
        elif in_asm and toks[0].startswith("global_load"):
            out.append(("aload", s, _regs(toks[1])))
        elif in_asm and toks[0].startswith("global_store"):
            out.append(("astore", s, regs))
        elif in_asm and toks[0] == "s_waitcnt" and "vmcnt" in s:
            out.append(("await", s, set()))
        else:
            out.append(("ins", s, regs))
    return out


def _walk(lines):
    """In-order model over prologue + main loop + main loop again: asserts that no compiler-generated instruction mentions
    the destination registers of an untracked load that is still counted as in flight.  Returns (parsed program, loop
    start, loop end, loads seen, waits seen)."""
    prog = _parse(lines)
    labels = {t: i for i, (k, t, _) in enumerate(prog) if k == "label"}
    best = None            # the main loop: the backward branch that spans the most untracked loads
    for i, (k, t, _) in enumerate(prog):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t) if k == "ins" else None
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < i:
                n = sum(1 for kk, _, _ in prog[tgt:i] if kk == "aload")
                if best is None or n > best[0]:
                    best = (n, tgt, i)
    assert best and best[0] > 0, "no loop with untracked loads found"
    _, lo, hi = best
    seq = prog[:hi + 1] + prog[lo:hi + 1]          # a load at the end of the body is awaited at its top
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
    # An asm store of more than 8 bytes keeps reading its data registers for two wait states after it issues (gfx940+ VMEM
    # store-data hazard; the compiler pads its own stores, not an asm statement's): whatever follows an untracked store
    # within two issue slots must be another store, a wait / nop, or must not WRITE the data registers.
    for i, (kind, text, regs) in enumerate(prog):
        if kind != "astore":
            continue
        data = _regs(re.split(r"[\s,]+", text)[2])
        slots = 0
        for k2, t2, r2 in prog[i + 1:]:
            if k2 == "label":
                continue
            op = re.split(r"[\s,]+", t2)
            if op[0] == "s_nop":
                slots += int(op[1]) + 1
            elif k2 in ("astore", "aload", "await") or op[0].startswith("s_"):
                slots += 1
            else:
                dest = _regs(op[1]) if len(op) > 1 else set()
                assert not (dest & data), f"`{t2}` writes v{sorted(dest & data)} {slots} wait state(s) behind `{text}`"
                slots += 1
            if slots >= 2:
                break
    return prog, lo, hi, n_loads, n_waits


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("nht,rb", [(16, 1), (8, 1), (4, 1), (2, 1), (8, 2)])
def test_no_instruction_touches_an_untracked_load_destination_before_the_hand_counted_wait(nht, rb):
    """Every width the chained kernel is built for (round 4: 16, 8, 4, 2 unit tiles = widths padding to 512, 256, 128, 64; a
    wave then carries 1, 1, 2 or 4 loader roles of 3 small loads each; with 16 unit tiles the loop body holds four sub-steps
    of 12 loads, two of them followed by the small loads)."""
    small = 3 * (((4 * rb + 3) * max(8 // nht, 1) + 7) // 8)      # (8, 2): two row blocks per minibatch, 11 loader roles
    prog, lo, hi, n_loads, n_waits = _walk(_kernel_asm(nht, rb))
    assert sum(1 for k, _, _ in prog[lo:hi + 1] if k == "aload") >= 2 * (12 + small), f"main loop with its 2 x (12 + {small}) loads not found"
    assert n_loads >= 3 * (12 + small) and n_waits >= 4
    # the loop's waits are the pipelined ones, and the loop is drained before the epilogue reuses registers
    in_loop = [t for k, t, _ in prog[lo:hi + 1] if k == "await"]
    assert in_loop and all("vmcnt(12)" in t for t in in_loop), in_loop
    after = [t for k, t, _ in prog[hi + 1:] if k == "await"]
    assert any("vmcnt(0)" in t for t in after), "no drain of the untracked loads after the loop"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("source,pattern,n_kernels", [("l1_gemm_i8.hip", r"^_Z17l1_gemm_i8_kernelILi", 6),
                                                      ("l1_gemm.hip", r"^_Z14l1_gemm_kernelILi", 3)])
def test_large_m_gemm_kernels_keep_their_untracked_fragment_loads_untouched_until_counted(source, pattern, n_kernels):
    """The same property for the hand-counted weight-fragment loads of the many-row layer-1 GEMMs (l1_gemm_i8.hip,
    l1_gemm.hip): every instantiation the library launches."""
    ks = _kernels(source, pattern)
    assert len(ks) == n_kernels, sorted(ks)
    for name, lines in ks.items():
        prog, lo, hi, n_loads, n_waits = _walk(lines)
        assert n_loads >= 16 and n_waits >= 8, (name, n_loads, n_waits)
