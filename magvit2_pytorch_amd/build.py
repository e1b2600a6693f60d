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
# product library: no environment switches, no experimental kernels
SOURCES = ['conv.hip', 'conv3.hip', 'conv3p.hip', 'convout.hip', 'convin.hip', 'pwse.hip', 'ff.hip', 'misc.hip', 'attn.hip', 'quant.hip', 'pack_host.hip']
# experiments library (tools/ only, -DMV2_EXPERIMENTS): the same sources with phase-clock profiling and the conv3 work-split knobs
LIB_EXP = os.path.join(HERE, 'libmv2_hip_exp.so')
# + kernels that exist only to be measured: they live with the other experiments under tools/experiments/, not in the product source directory
EXP_DIR = os.path.join(HERE, '..', 'tools', 'experiments')
SOURCES_EXP = SOURCES + [os.path.join(EXP_DIR, 'exp_se.hip'), os.path.join(EXP_DIR, 'exp_attn.hip')]
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_common.h'), os.path.join(CSRC, 'conv3p.h'), os.path.join(CSRC, 'conv3_knobs.h'),
           os.path.join(HERE, '..', 'include', 'mv2.h')]
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


def build(force=False, verbose=True, experiments=False):
    """experiments=True builds libmv2_hip_exp.so (the A/B library of tools/) instead of the product library."""
    sources, lib, objdir = (SOURCES_EXP, LIB_EXP, OBJ + '_exp') if experiments else (SOURCES, LIB, OBJ)
    flags = FLAGS + (['-DMV2_EXPERIMENTS'] if experiments else [])
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in sources:
        src = s if os.path.isabs(s) else os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.basename(s).replace('.hip', '.o'))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *flags, '-I', CSRC, '-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
        return obj

    if jobs:
        if verbose:
            print(f'[mv2 build] compiling {len(jobs)} file(s) for gfx950 ...', flush=True)
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, os.path.basename(s).replace('.hip', '.o')) for s in sources]
    if force or jobs or _stale(lib, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'[mv2 build] linked {lib}', flush=True)
    return lib


if __name__ == '__main__':
    build(force='--force' in sys.argv, experiments='--experiments' in sys.argv)
