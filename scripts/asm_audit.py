"""Audit of a kernel whose LDS reads are inline asm (gemm256n4.hip): hipcc treats the destination of an `asm("ds_read_b128 ..")` as written when the
statement ends, so under register pressure it may COPY, SPILL or REUSE those registers before the data has landed - garbage on some waves of some
launches, and nothing in a passing test says so (cdna_hip_programming.md 5.7 item 1).  This script replays the kernel's listing:

  * every asm ds_read destination is "pending" until an s_waitcnt lgkmcnt(N) has retired it (LDS returns in order: the N youngest stay pending);
  * any instruction OUTSIDE an asm statement that reads or writes a pending register is reported (within a basic block: the replay restarts at
    labels and branches);
  * MFMAs written as asm are unknown to the compiler's hazard padding: a VALU write of an MFMA source operand less than 2 instructions ahead of the
    MFMA is reported as well (v_mov / v_accvgpr copies of fragments the allocator inserts);
  * scratch accesses (spills) and v_accvgpr_write (accumulators migrating between the register files) are counted per kernel.

    python scripts/asm_audit.py leccr_amd/csrc/gemm256n4.hip [kernel-name-substring] [-DFOO=1 ...]
Needs hipcc (cross-compiles here, no GPU).  Exit code 1 when a pending register is touched."""
import os, re, subprocess, sys, tempfile

args = sys.argv[1:]
src = args[0]
want = next((a for a in args[1:] if not a.startswith('-')), '')
defs = [a for a in args[1:] if a.startswith('-D')]
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, 'k.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', asm, src] + defs,
                   check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read()


def regs(tok):
    """v[a:b] / vN / a[..] -> set of ('v', n)"""
    out = set()
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]', tok):
        out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r'\b([va])(\d+)\b', tok):
        out.add((m.group(1), int(m.group(2))))
    return out


bad = 0
for km in re.finditer(r'^(_Z\w+):\s*; @', text, re.M):
    name = km.group(1)
    if want not in name:
        continue
    body = text[km.start():text.find('s_endpgm', km.start())].split('\n')
    pending = []          # FIFO of (line, set of regs)
    in_asm = False
    recent_valu = []      # (line no, written regs) of the last compiler VALU writes
    n_scratch = n_accw = n_mfma = carried = 0
    findings = []
    for ln, raw in enumerate(body):
        l = raw.strip()
        if l.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if l.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if l.startswith('.LBB') or (l.split() and (l.split()[0].startswith('s_cbranch') or l.split()[0].startswith('s_branch'))):
            # a basic-block boundary: the listing's order is not the program's (hipcc lays the blocks of a K-tile body out where it likes), so the
            # replay starts afresh - reads still pending here are counted, not followed
            carried += len(pending)
            pending = []
            continue
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        op = l.split()[0]
        operands = l[len(op):].split(';')[0]
        parts = [p.strip() for p in operands.split(',')]
        if op.startswith('scratch_'): n_scratch += 1
        if op.startswith('v_accvgpr_write'): n_accw += 1
        if op.startswith('s_waitcnt'):
            m = re.search(r'lgkmcnt\((\d+)\)', l)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
            elif 'vmcnt' not in l and 'expcnt' not in l:      # numeric form: treat as a full wait
                pending = []
            continue
        if op.startswith('ds_read') and in_asm:
            pending.append((ln, regs(parts[0])))
            continue
        if op.startswith('v_mfma'):
            n_mfma += 1
            srcs = regs(parts[1]) | regs(parts[2])
            for pl, pr in pending:
                if srcs & pr:
                    findings.append(f'  line {ln}: MFMA reads {sorted(srcs & pr)[:4]} still pending from the ds_read at line {pl}: {l}')
            for vl, vr in recent_valu:
                if ln - vl <= 2 and srcs & vr:
                    findings.append(f'  line {ln}: MFMA source written by a compiler VALU instruction {ln - vl} lines earlier (no hazard padding for asm): {l}')
            continue
        if in_asm:
            continue
        touched = regs(operands)
        for pl, pr in pending:
            if touched & pr:
                findings.append(f'  line {ln}: compiler instruction touches {sorted(touched & pr)[:4]} pending from the ds_read at line {pl}: {l}')
        if op.startswith('v_') and parts:
            recent_valu.append((ln, regs(parts[0])))
            recent_valu = recent_valu[-4:]
    print(f'{name}: {n_mfma} MFMAs, {n_scratch} scratch accesses, {n_accw} v_accvgpr_write, {carried} reads pending across block boundaries (not followed), {len(findings)} findings')
    for f in findings[:40]:
        print(f)
    bad += len(findings)
sys.exit(1 if bad else 0)
