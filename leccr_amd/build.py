"""Build libleccr_hip.so (every HIP kernel + the C ABI) for gfx950 with hipcc, in-tree.

    python -m leccr_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached under leccr_amd/csrc/build/ keyed by source mtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libleccr_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Wno-unused-result']
# Per-source extras.  attention_mfma: the SLP vectoriser pairs the f32 multiplies of the softmax / dS stages into v_pk_mul_f32
# and glues the bf16 packs together with v_pk_mov / v_alignbit / v_perm (50 extra VALU instructions per tile in kernels the VALU
# bounds; packed f32 ops also issue at half rate beside MFMAs): backward -3..5 % without it.
EXTRA_FLAGS = {'attention_mfma.hip': ['-fno-slp-vectorize']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(src, obj, headers):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in [src] + headers)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'leccr_hip.h'))
    jobs = []
    objs = []
    for s in sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + '.o')
        objs.append(obj)
        if force or _stale(src, obj, headers):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB) or force:
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
