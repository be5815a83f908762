"""Loop "signatures" of the hand-scheduled kernels as hipcc emitted them: one letter per instruction class in program order, for every loop of a
kernel that contains MFMAs - enough to see at a glance whether the interleave written in the source (MFMA i, then fill i; sched_barrier fences)
survived the compiler.  Round 4 found two fp8 kernels where it had not: a branch between two k-steps ended the basic block and hipcc's sinking pass
moved a k-step's 16 MFMAs below all of its fragment reads (profiles/r04_fp8_kstep_schedules.txt); sched_barrier does not bind that pass.

    python scripts/kernel_listing.py leccr_amd/csrc/gemm256t4.hip [kernel-name-substring] [--opcodes OUT.txt]

  M  v_mfma_*            d  ds_read*          w  ds_write*         G  LDS-DMA (global_load_lds / buffer_load .. lds)
  g  global_load         s  global_store / atomic                  S  scratch (a spill inside a loop is a finding)
  e  v_exp               c  v_dot2*           .  other VALU        B  s_barrier        _  s_waitcnt        |  branch        :  label

--opcodes writes the bare opcode sequence of every kernel (labels dropped): `diff` of two such files tells whether an edit changed the code of
a kernel it was not meant to touch (how the T4_SCHED shadowing of round 4 was found).
Needs hipcc (cross-compiles here, no GPU)."""
import os, re, subprocess, sys, tempfile

args = [a for a in sys.argv[1:] if not a.startswith('--')]
src, want = args[0], (args[1] if len(args) > 1 else '')
opc_out = sys.argv[sys.argv.index('--opcodes') + 1] if '--opcodes' in sys.argv else None
extra = ['-fno-slp-vectorize'] if os.path.basename(src) == 'attention_mfma.hip' else []          # as leccr_amd/build.py builds it
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, 'k.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', asm, src] + extra,
                   check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read()


def letter(t):
    op = t.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_read'): return 'd'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'w'
    if 'global_load_lds' in op or (op.startswith('buffer_load') and ' lds' in t): return 'G'
    if op.startswith('global_load'): return 'g'
    if op.startswith('global_store') or op.startswith('global_atomic'): return 's'
    if op.startswith('scratch_'): return 'S'
    if op.startswith('v_exp'): return 'e'
    if op.startswith('v_dot2'): return 'c'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return '|'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith('s_waitcnt'): return '_'
    if op.startswith('v_'): return '.'
    return ''


opcodes = []
for km in re.finditer(r'^(_Z\w+):\s*; @', text, re.M):
    name = km.group(1)
    body = text[km.start():text.find('s_endpgm', km.start())].split('\n')
    if opc_out:
        opcodes.append('== ' + name)
        opcodes += [l.split()[0] for l in (x.strip() for x in body[1:]) if l and not l.startswith(';') and not l.startswith('.')]
    if want not in name:
        continue
    labels = {m.group(1): n for n, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    print(name)
    for n, l in enumerate(body):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:           # a backward branch: [label, here] is a loop
            seg = body[labels[m.group(1)]:n + 1]
            if not any('v_mfma' in x for x in seg):
                continue
            sig = ''.join(':' if re.match(r'^\.LBB', t) else letter(t) for t in (x.strip() for x in seg) if t and not t.startswith(';'))
            print(f'  loop {m.group(1)}: {len(seg)} lines, {sig.count("M")} MFMAs, {sig.count("S")} scratch')
            print('    ' + sig)
if opc_out:
    open(opc_out, 'w').write('\n'.join(opcodes) + '\n')
