"""The library's host pass under AddressSanitizer + UndefinedBehaviorSanitizer ("sanitizer build"; the GPU-side
sanitizers are not available on this pool).  CPU only: lime_cikm25_amd.build.build_sanitized() compiles build/liblime_hip_san.so, and a
C driver GENERATED here from the binding table (lime_cikm25_amd._lib.SIGNATURES: the same list tests/test_abi.py holds against
include/lime_hip.h), built with the same sanitizers, calls every entry point with NULL / zero arguments -- argument validation has to
reject them before anything is launched --, every args struct zeroed and with dimensions but no pointers, the size functions at real
and at extreme dimensions, and reads the error string.  A heap / stack error or undefined behaviour in that host code (validation,
dispatch, launch geometry, workspace sizing) ends the driver with the sanitizer's report."""
import ctypes
import os
import shutil
import subprocess

import pytest

from lime_cikm25_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_ZERO_CALL = ('lime_set_split_gemm',)


def _driver_source():
    lines = ['#include <stdio.h>', '#include <string.h>', '#include "%s"' % os.path.join(ROOT, 'include', 'lime_hip.h'),
             '#define CHECK(x) do { if (!(x)) { printf("FAILED: %s\\n", #x); return 1; } } while (0)', 'int main(void) {', '  int n = 0;']
    for name, (res, args) in sorted(_lib.SIGNATURES.items()):
        if name in NO_ZERO_CALL:
            continue
        call = '%s(%s)' % (name, ', '.join('0' for _ in args))
        is_status = res is ctypes.c_int32 and not name.endswith(('_workspace', '_size', '_columns', '_version'))
        lines.append('  { long long r = (long long)%s; n++; %s }' % (call, 'CHECK(r <= 0);' if is_status else '(void)r;'))
    for cname, entry, dims in (('lime_linear_args', 'lime_linear_f32', ('M', 'N', 'K')), ('lime_linear_bf16_args', 'lime_linear_bf16', ('M', 'N', 'K')),
                               ('lime_ffn_bf16_args', 'lime_encoder_ffn_bf16', ('M', 'E', 'F')),
                               ('lime_encoder_block_bf16_args', 'lime_encoder_block_bf16', ('M', 'E', 'F')),
                               ('lime_inproj_bf16_args', 'lime_inproj_bf16', ('M', 'N', 'K'))):
        lines += ['  { %s a; memset(&a, 0, sizeof a); CHECK(%s(&a, 0) < 0);' % (cname, entry)] + \
                 ['    a.%s = 4096;' % d for d in dims] + ['    CHECK(%s(&a, 0) < 0); CHECK(strlen(lime_last_error_string()) > 0); n += 2; }' % entry]
    lines += ['  CHECK(lime_token_attention_bwd_workspace(1760, 512, 10) > lime_token_attention_stats_workspace(1760, 512, 10));',
              '  CHECK(lime_token_attention_bwd_workspace(1 << 20, 512, 16) > 0);      /* 64-bit arithmetic: no signed overflow */',
              '  CHECK(lime_token_attention_bwd_workspace(0, 32, 1) == 0);',
              '  CHECK(lime_cand_attn_weights_workspace(1 << 15, 128, 512, 16) > 0);',
              '  CHECK(lime_compact_sequences_workspace(1 << 24) > 0);',
              '  CHECK(lime_layernorm_bwd_workspace(1 << 24, 512) > 0);',
              '  CHECK(lime_set_split_gemm(-1) >= 0);',
              '  CHECK(lime_abi_version() == LIME_ABI_VERSION);',
              '  printf("sanitizer driver: %d calls\\n", n);', '  return 0;', '}']
    return '\n'.join(lines) + '\n'


def test_host_pass_is_clean_under_asan_and_ubsan(tmp_path):
    clang = '/opt/rocm/lib/llvm/bin/clang'
    if not os.path.exists(clang) or shutil.which('hipcc') is None:
        pytest.skip('no hipcc / clang')
    lib = os.path.abspath(build.build_sanitized())
    src = tmp_path / 'san_driver.c'
    src.write_text(_driver_source())
    exe = tmp_path / 'san_driver'
    r = subprocess.run([clang, '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-o', str(exe), str(src), lib,
                        '-Wl,-rpath,' + os.path.dirname(lib), '-Wl,-rpath,/opt/rocm/lib'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run(['timeout', '-s', 'KILL', '120', str(exe)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, 'sanitizer driver failed (rc %d):\n%s\n%s' % (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    assert 'sanitizer driver:' in r.stdout and 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-4000:]
