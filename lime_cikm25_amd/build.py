"""Build liblime_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB = os.path.join(_HERE, 'liblime_hip.so')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(_HERE, '..', 'include', 'lime_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source into lime_cikm25_amd/liblime_hip.so; returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build the HIP extension')
    cmd = [hipcc] + FLAGS + ['-o', LIB + '.tmp'] + sources()
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
