"""
Builds libmv2_hip.so (the C-ABI of include/mv2.h) in-tree with hipcc for gfx950.

    python -m magvit2_pytorch_amd.build [--force]

hipcc cross-compiles without a GPU; the .so stays in the package directory (git-ignored, but it
travels with the gpurun snapshot) so GPU boxes never need to compile.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', '_obj')
LIB = os.path.join(HERE, 'libmv2_hip.so')
SOURCES = ['conv.hip', 'conv_halo.hip', 'conv_halo_rs.hip', 'misc.hip', 'attn.hip', 'quant.hip']
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_common.h'), os.path.join(HERE, '..', 'include', 'mv2.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace('.hip', '.o'))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, '-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
        return obj

    if jobs:
        if verbose:
            print(f'[mv2 build] compiling {len(jobs)} file(s) for gfx950 ...', flush=True)
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'[mv2 build] linked {LIB}', flush=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
