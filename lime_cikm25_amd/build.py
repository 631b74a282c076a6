"""Build liblime_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is rebuilt when the CONTENT of its sources changes: a sha256 over every file under csrc/, include/lime_hip.h and
the compiler flags is stored beside the binary (liblime_hip.so.sha256, which travels with it); file times play no part, so a
stale-but-newer binary is never reused.  ``build_library`` reports which of the two happened."""
import glob
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB = os.path.join(_HERE, 'liblime_hip.so')
STAMP = LIB + '.sha256'
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared']
LAST_ACTION = None            # 'compiled' | 'reused' after build_library()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def source_hash():
    """sha256 over (relative name, content) of csrc/*.{hip,cpp,h} and include/lime_hip.h, and the flags."""
    h = hashlib.sha256(' '.join(FLAGS).encode())
    deps = sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(_HERE, '..', 'include', 'lime_hip.h')]
    for d in deps:
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp():
    try:
        return open(STAMP).read().strip()
    except OSError:
        return None


def build_library(force=False, verbose=False):
    """Compile every HIP source into lime_cikm25_amd/liblime_hip.so; returns the path.  Sets LAST_ACTION and prints one line
    saying whether the library was compiled or reused, with the source hash."""
    global LAST_ACTION
    want = source_hash()
    if not force and os.path.exists(LIB) and _stamp() == want:
        LAST_ACTION = 'reused'
        print('liblime_hip.so: reused (csrc sha256 %s matches the stamp beside the binary)' % want[:16])
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build the HIP extension')
    cmd = [hipcc] + FLAGS + ['-o', LIB + '.tmp'] + sources()
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + '.tmp', LIB)
    with open(STAMP, 'w') as f:
        f.write(want + '\n')
    LAST_ACTION = 'compiled'
    print('liblime_hip.so: compiled (csrc sha256 %s, %d bytes)' % (want[:16], os.path.getsize(LIB)))
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
