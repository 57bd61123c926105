"""Builds libclamd.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python continual-learning_amd/build.py [--force]
    python continual-learning_amd/build.py --variant NAME [-DFLAG ...] [--diag]     # experiment build -> build/NAME/libclamd.so (CLAMD_LIB)

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libclamd.so')
SOURCES = ['igemm.hip', 'igemm_ws.hip', 'igemm_pws.hip', 'wgrad.hip', 'wgrad_dma.hip', 'wino.hip', 'wino24.hip', 'wino24n.hip', 'wino24_wgrad.hip', 'wino24g.hip', 'wino44g.hip', 'bnfold.hip', 'elementwise.hip', 'misc.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-I' + os.path.join(HERE, '..', 'include')]


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, '..', 'include')):
        for f in sorted(os.listdir(root)):
            if f.endswith(('.hip', '.h')):
                h.update(f.encode())
                h.update(open(os.path.join(root, f), 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True, diag=False, variant=None, extra_flags=()):
    """variant: name of an EXPERIMENT build (A/B and diagnostic tools): written to build/<variant>/libclamd.so with its own objects,
    loaded by a process that sets CLAMD_LIB to that path; the product library (variant None) is never touched by it."""
    global FLAGS
    flags = list(FLAGS)
    if os.environ.get('CLAMD_EXTRA_FLAGS'):     # experiments (ablation builds): never for measurements that are reported
        flags += os.environ['CLAMD_EXTRA_FLAGS'].split()
    flags += list(extra_flags)
    if diag:
        flags += ['-DCLAMD_DIAG']             # diagnostic build: in-kernel cycle stamps (never for measurements)
    out, objdir = OUT, CSRC
    if variant:
        objdir = os.path.join(HERE, '..', 'build', variant)
        os.makedirs(objdir, exist_ok=True)
        out = os.path.join(objdir, 'libclamd.so')
    stamp = os.path.join(objdir, '.build_stamp')
    saved, FLAGS = FLAGS, flags
    dig = _digest()
    FLAGS = saved
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig:
        return out
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    # objects of sources that have left the library (an experiment removed again) must not travel to the GPU box with the snapshot
    keep = {src.replace('.hip', '.o') for src in SOURCES}
    for f in os.listdir(objdir):
        if f.endswith('.o') and f not in keep:
            os.remove(os.path.join(objdir, f))

    def cc(src):
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        cmd = [hipcc] + flags + ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{r.stderr[-6000:]}')
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr[-3000:])
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    with open(stamp, 'w') as f:
        f.write(dig)
    return out


if __name__ == '__main__':
    # python build.py [--force] [--diag] [--variant NAME [-Dflag ...]]
    var = sys.argv[sys.argv.index('--variant') + 1] if '--variant' in sys.argv else None
    print(build(force='--force' in sys.argv or ('--diag' in sys.argv and not var), diag='--diag' in sys.argv, variant=var,
                extra_flags=[a for a in sys.argv[1:] if a.startswith('-D')]))
