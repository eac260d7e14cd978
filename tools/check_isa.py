#!/usr/bin/env python3
"""ISA lint for the kernels that issue VMEM instructions from inline asm (gemm256sp.hip, gemm256w4.hip).

hipcc's hazard recognizer does not look inside inline asm, so two gfx9 hazards have to be kept away by construction:
  * an SGPR written by a VALU (v_readlane / v_readfirstlane, e.g. the reload of a spilled SGPR) must not be read by a
    VMEM instruction within the next 5 wait states;
  * no VGPR spills: a scratch reload carries `s_waitcnt vmcnt(0)`, which also waits for every store before it.
Usage: check_isa.py file.hip [...]   (exit status 1 on a finding)
"""
import re
import subprocess
import sys
import tempfile

VMEM = ("global_", "buffer_", "scratch_", "flat_")


def sgprs(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-o", f.name, path], check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    findings = []
    for m in re.finditer(r"\.name:\s+(\S+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S):
        # EPI 4 (patch embedding: once per forward) always spilled; EPI 3 (the f32 residual epilogue: a test hook since the residual
        # stream moved to fp16 in round 3, not launched by the encoder) spills 6 registers since the 16x16x32 form (round 4)
        if "gemm256w4_kernel" in m.group(1) and int(m.group(2)):
            findings.append(f"{m.group(1)}: {m.group(2)} VGPR spills")
        if "gemm256sp_kernel" in m.group(1) and "ILi4E" not in m.group(1) and "ILi3E" not in m.group(1) and int(m.group(2)):
            findings.append(f"{m.group(1)}: {m.group(2)} VGPR spills")
    ins = [l.strip() for l in text.split("\n")]
    ins = [l for l in ins if l and not l.startswith((";", ".")) and not l.endswith(":")]
    for i, l in enumerate(ins):
        if not l.startswith(("v_readlane_b32", "v_readfirstlane_b32")):
            continue
        dst = sgprs(l.split()[1].rstrip(","))
        states = 0
        for nxt in ins[i + 1:i + 8]:
            if nxt.startswith(VMEM):
                used = set()
                for tok in re.findall(r"s\[\d+:\d+\]|\bs\d+\b", nxt):
                    used |= sgprs(tok)
                if dst & used and states < 5:
                    findings.append(f"VALU-written SGPR read by VMEM after {states} wait states: {l}  ->  {nxt}")
            states += (int(nxt.split()[1]) + 1) if nxt.startswith("s_nop") else 1
            if states >= 5:
                break
    findings += check_pending_lds_reads(text)
    return findings


def vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check_pending_lds_reads(text):
    """An inline-asm ds_read's destination counts as written at the end of the asm statement, so hipcc may touch it before
    the data lands.  Replay every kernel linearly (no control-flow graph: restart after unconditional branches) with a FIFO
    of outstanding LDS operations (LDS returns in order;
    `s_waitcnt lgkmcnt(N)` leaves the N youngest) and flag any instruction OUTSIDE an asm statement that names a VGPR an
    asm ds_read still has in flight (a v_mov / spill / early MFMA use of a fragment register)."""
    findings = []
    kernel, in_asm, fifo = None, False, []
    for raw in text.split("\n"):
        l = raw.strip()
        m = re.match(r"(_Z\S+):", raw)
        if m:
            kernel, fifo = m.group(1), []
            continue
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith((";", ".")) or l.endswith(":") or kernel is None or "gemm" not in kernel:
            continue
        op = l.split()[0]
        if op == "s_endpgm":
            kernel = None
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                fifo = fifo[len(fifo) - n:] if n else []
            continue
        if op == "s_barrier":
            continue
        if op == "s_branch":  # what follows is reached from elsewhere: the replay is linear, so start over
            fifo = []
            continue
        if op.startswith(("ds_", "s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            dst = set()
            if in_asm and op.startswith("ds_read"):
                dst = vregs(l.split()[1].rstrip(","))
            fifo.append(dst)
            if not in_asm:  # a compiler LDS op may of course name its own registers; only asm destinations are tracked
                continue
        if in_asm:
            continue
        pending = set().union(*fifo) if fifo else set()
        if not pending:
            continue
        used = set()
        for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", l):
            used |= vregs(tok)
        hit = used & pending
        if hit:
            findings.append(f"{kernel[:60]}: `{l}` touches v{sorted(hit)[0]}.. while an inline-asm ds_read into it is still in flight")
    return findings[:20]


if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        bad += [f"{p}: {x}" for x in check(p)]
    print("\n".join(bad) if bad else "ISA lint: clean")
    sys.exit(1 if bad else 0)
