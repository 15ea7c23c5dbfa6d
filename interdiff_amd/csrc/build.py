"""Build libinterdiff_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m interdiff_amd.csrc.build [--force]

One translation unit per .hip file, objects cached next to the sources (git-ignored), linked
into interdiff_amd/csrc/libinterdiff_hip.so -- in-tree so that it travels to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libinterdiff_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Wno-inline-asm',        # the asm LDS-DMA names m0 as a clobber on purpose (csrc/common.h idf_dma16_*)
         # gfx950 kernel-argument preloading: the leading arguments of every kernel (as many as fit the 14 free user SGPRs) are in registers when a wave
         # starts instead of behind an s_load round trip to the argument segment -- 22 kernel starts per denoising step; same-box A/B of the whole library
         # -1.0 % per step (profiles/r05_lib_ab_preload.txt).  Kernels whose first loads matter order their arguments for it (denoiser.hip rowblock8_kernel).
         '-mllvm', '-amdgpu-kernarg-preload-count=16']
FLAGS += os.environ.get('IDF_EXTRA_HIPCC_FLAGS', '').split()      # A/B builds (e.g. -DIDF_WT_MODE=2); empty for the product


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out_dir=None):
    """out_dir (or IDF_BUILD_DIR): objects and the library go THERE instead of next to the sources -- variant builds for A/B tools (build_ab/<name>/,
    loaded through INTERDIFF_HIP_LIB); the product library is always the in-tree one."""
    out_dir = out_dir or os.environ.get('IDF_BUILD_DIR') or HERE
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, 'libinterdiff_hip.so')
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')]
    hdrs.append(os.path.join(HERE, '..', '..', 'include', 'interdiff_hip.h'))
    objs, jobs = [], []
    for s in sources():
        src, obj = os.path.join(HERE, s), os.path.join(out_dir, s[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
