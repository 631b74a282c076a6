"""Build liblime_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object file per translation unit (build/obj, up to 7 hipcc processes at once; a unit is recompiled when its own text, a
csrc header or the flags changed), then one link.  The library is rebuilt when the CONTENT of its sources changes: a sha256 over every file under csrc/, include/lime_hip.h and
the compiler flags is stored beside the binary (liblime_hip.so.sha256, which travels with it); file times play no part, so a
stale-but-newer binary is never reused.  ``build_library`` reports which of the two happened."""
import glob
import hashlib
import os
import json
import re
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB = os.path.join(_HERE, 'liblime_hip.so')
STAMP = LIB + '.sha256'
OBJ_DIR = os.path.join(_HERE, '..', 'build', 'obj')
RESOURCES = os.path.join(_HERE, 'liblime_hip.resources.json')
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


def _object_key(src, extra_flags=()):
    """sha256 of one translation unit's inputs: its own text, every csrc/*.h, include/lime_hip.h, the flags."""
    h = hashlib.sha256(' '.join(list(FLAGS) + list(extra_flags)).encode())
    for d in [src] + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(_HERE, '..', 'include', 'lime_hip.h')]:
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def compile_objects(verbose=False, jobs=None):
    """One object per source under build/obj (git-ignored, never shipped), recompiled only when that unit's inputs changed;
    up to `jobs` hipcc processes at once.  Returns the object paths in source order."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build the HIP extension')
    os.makedirs(OBJ_DIR, exist_ok=True)
    cflags = [f for f in FLAGS if f != '-shared']
    todo, objs = [], []
    for src in sources():
        stem = os.path.splitext(os.path.basename(src))[0]
        obj, key = os.path.join(OBJ_DIR, stem + '.o'), _object_key(src)
        objs.append(obj)
        try:
            fresh = os.path.exists(obj) and open(obj + '.key').read().strip() == key
        except OSError:
            fresh = False
        if not fresh:
            todo.append((src, obj, key))
    jobs = jobs or max(1, min(len(todo), (os.cpu_count() or 2) - 1, 7))
    running = []

    def reap(block):
        for item in list(running):
            proc, src, obj, key = item
            if block or proc.poll() is not None:
                if proc.wait() != 0:
                    sys.stderr.write(open(obj + '.rpass').read()[-4000:])
                    for other in running:
                        if other[0].poll() is None:
                            other[0].kill()
                    raise RuntimeError('hipcc failed on %s' % os.path.basename(src))
                with open(obj + '.key', 'w') as f:
                    f.write(key + '\n')
                running.remove(item)
                if block:
                    return

    # longest units first, so the tail of the schedule is short
    for src, obj, key in sorted(todo, key=lambda t: -os.path.getsize(t[0])):
        while len(running) >= jobs:
            reap(False)
            if len(running) >= jobs:
                reap(True)
        # -Rpass-analysis prints each kernel's registers / scratch / LDS as remarks (no effect on the code): kept beside the
        # object, summarised into liblime_hip.resources.json (tests/test_kernel_resources.py reads it)
        cmd = [hipcc] + cflags + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        running.append((subprocess.Popen(cmd, cwd=CSRC, stderr=open(obj + '.rpass', 'w')), src, obj, key))
    while running:
        reap(True)
    return objs, len(todo)


def _demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return [o.replace('(anonymous namespace)::', '').replace('void ', '') for o in out[:len(names)]]


def kernel_resources(objs):
    """{unit: {kernel: {vgprs, agprs, sgprs, vgpr_spill, sgpr_spill, scratch, lds, occupancy}}} from the remarks hipcc printed
    while compiling each unit."""
    res = {}
    for obj in objs:
        unit, rows, cur = os.path.splitext(os.path.basename(obj))[0], [], None
        for line in open(obj + '.rpass', errors='replace'):
            m = re.search(r'remark: +(?:\[[^\]]*\] *)?(.*?): +(\S+)\s*(?:\[-Rpass|$)', line)
            if not m:
                continue
            k, v = m.group(1).strip(), m.group(2)
            if k == 'Function Name':
                cur = {'name': v}
                rows.append(cur)
            elif cur is not None:
                cur[k] = v
        names = _demangle([r['name'] for r in rows]) if rows else []
        num = lambda r, k: int(r.get(k, '0') or 0)
        res[unit] = {n: {'vgprs': num(r, 'VGPRs'), 'agprs': num(r, 'AGPRs'), 'sgprs': num(r, 'SGPRs'), 'vgpr_spill': num(r, 'VGPRs Spill'),
                         'sgpr_spill': num(r, 'SGPRs Spill'), 'scratch': num(r, 'ScratchSize [bytes/lane]'),
                         'lds': num(r, 'LDS Size [bytes/block]'), 'occupancy': num(r, 'Occupancy [waves/SIMD]')}
                     for n, r in zip(names, rows)}
    return res


SAN_LIB = os.path.join(_HERE, '..', 'build', 'liblime_hip_san.so')
SAN_FLAGS = ['-O1', '-g', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fsanitize=address,undefined', '-fno-omit-frame-pointer',
             '-fno-sanitize-recover=undefined', '-Wno-option-ignored']


def build_sanitized(verbose=False):
    """The library's HOST pass under AddressSanitizer + UndefinedBehaviorSanitizer (the sanitizer build of the
    reference's tooling; GPU sanitizers are not available on this pool): every unit recompiled with -fsanitize=address,undefined
    -- hipcc applies it to the host code (argument validation, dispatch, launch geometry, workspace sizing) and ignores it for the
    gfx950 pass -- into build/liblime_hip_san.so.  tests/test_sanitizers.py drives it without a GPU, from an instrumented C driver (a python with
    the ASan runtime preloaded does not start in this container)."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    objdir = os.path.join(_HERE, '..', 'build', 'obj_san')
    os.makedirs(objdir, exist_ok=True)
    want = hashlib.sha256((source_hash() + ' '.join(SAN_FLAGS)).encode()).hexdigest()
    stamp = SAN_LIB + '.sha256'
    try:
        if os.path.exists(SAN_LIB) and open(stamp).read().strip() == want:
            return SAN_LIB
    except OSError:
        pass
    procs, objs = [], []
    jobs = max(1, min((os.cpu_count() or 2) - 1, 7))
    for src in sorted(sources(), key=lambda p: -os.path.getsize(p)):
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + '.o')
        objs.append(obj)
        while len([p for p in procs if p.poll() is None]) >= jobs:
            [p for p in procs if p.poll() is None][0].wait()
        cmd = [hipcc] + SAN_FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append(subprocess.Popen(cmd, cwd=CSRC, stderr=subprocess.DEVNULL))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError('sanitized compile failed')
    # the sanitizer runtime itself comes from the (instrumented) executable that loads the library
    subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-fsanitize=address,undefined', '-o', SAN_LIB] + objs,
                   check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    with open(stamp, 'w') as f:
        f.write(want + '\n')
    return SAN_LIB


def build_library(force=False, verbose=False):
    """Compile every HIP source into lime_cikm25_amd/liblime_hip.so; returns the path.  Sets LAST_ACTION and prints one line
    saying whether the library was compiled or reused, with the source hash."""
    global LAST_ACTION
    want = source_hash()
    if not force and os.path.exists(LIB) and _stamp() == want:
        LAST_ACTION = 'reused'
        print('liblime_hip.so: reused (csrc sha256 %s matches the stamp beside the binary)' % want[:16])
        return LIB
    if force:
        shutil.rmtree(OBJ_DIR, ignore_errors=True)
    objs, n_compiled = compile_objects(verbose=verbose)
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB + '.tmp'] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + '.tmp', LIB)
    with open(RESOURCES, 'w') as f:
        json.dump({'source_hash': want, 'units': kernel_resources(objs)}, f, indent=1, sort_keys=True)
    with open(STAMP, 'w') as f:
        f.write(want + '\n')
    LAST_ACTION = 'compiled'
    print('liblime_hip.so: compiled (%d of %d units rebuilt, csrc sha256 %s, %d bytes)' % (n_compiled, len(objs), want[:16],
                                                                                        os.path.getsize(LIB)))
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
