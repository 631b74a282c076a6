"""The C-ABI shared library: loads, exports every symbol include/lime_hip.h declares, rejects bad
arguments before touching the GPU.  No compute calls here (CPU only)."""
import ctypes
import os
import re

import pytest

from lime_cikm25_amd import _lib
from lime_cikm25_amd.build import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    build_library()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'lime_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lime_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree(lib):
    names = declared_symbols()
    assert len(names) >= 15
    assert sorted(_lib.SIGNATURES) == names
    for n in names:
        assert getattr(lib, n) is not None


def test_abi_version(lib):
    header = open(os.path.join(ROOT, 'include', 'lime_hip.h')).read()
    declared = int(re.search(r'#define\s+LIME_ABI_VERSION\s+(\d+)', header).group(1))
    assert lib.lime_abi_version() == _lib.ABI_VERSION == declared


def _c_layout(tmp_path):
    """sizeof / offsetof of lime_linear_args as gcc sees the header."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu", '
                   'sizeof(lime_linear_args), offsetof(lime_linear_args, c), offsetof(lime_linear_args, act), '
                   'offsetof(lime_linear_args, ln_eps)); return 0;}' % os.path.join(ROOT, 'include', 'lime_hip.h'))
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', '-o', str(exe), str(src)], check=True)
    return tuple(int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())


def test_linear_args_layout_matches_header(tmp_path):
    # the ctypes mirror must have exactly the layout the C compiler gives the struct in include/lime_hip.h
    LAYOUT = _c_layout(tmp_path)
    assert ctypes.sizeof(_lib.LinearArgs) == LAYOUT[0]
    assert (_lib.LinearArgs.c.offset, _lib.LinearArgs.act.offset, _lib.LinearArgs.ln_eps.offset) == LAYOUT[1:]


@pytest.mark.parametrize('cname,cls', [('lime_linear_args', 'LinearArgs'), ('lime_linear_bf16_args', 'LinearBf16Args'),
                                        ('lime_ffn_bf16_args', 'FfnBf16Args'), ('lime_encoder_block_bf16_args', 'EncoderBlockBf16Args'),
                                        ('lime_inproj_bf16_args', 'InprojBf16Args')])
def test_every_args_struct_matches_the_header_field_by_field(tmp_path, cname, cls):
    """sizeof and the offset of EVERY field of the ctypes mirror against what gcc gives the struct of include/lime_hip.h."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    st = getattr(_lib, cls)
    fields = [f[0] for f in st._fields_]
    src = tmp_path / 'off.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu", sizeof(%s));\n%s\nreturn 0;}' % (
        os.path.join(ROOT, 'include', 'lime_hip.h'), cname,
        '\n'.join('printf(" %%zu", offsetof(%s, %s));' % (cname, f) for f in fields)))
    exe = tmp_path / 'off'
    subprocess.run(['gcc', '-o', str(exe), str(src)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert ctypes.sizeof(st) == got[0], cname
    assert [getattr(st, f).offset for f in fields] == got[1:], cname


def test_bad_arguments_are_rejected_without_a_launch(lib):
    assert lib.lime_linear_f32(None, None) == -1
    assert b'NULL' in lib.lime_last_error_string()
    a = _lib.LinearArgs()
    assert lib.lime_linear_f32(ctypes.byref(a), None) == -1
    assert lib.lime_bucketize_f32(None, None, 4, None) == -1
    assert lib.lime_token_attention_f32(None, None, None, 0, None, None, 0, 1, 1, 1, 1, 1, 1.0, None) == -1
    assert lib.lime_pad_heads_f32(None, 1, None, 1, 1, 1, 1, 1, None) == -1
    assert lib.lime_multi_copy(None, 3, None) == -1
    assert lib.lime_multi_copy(None, 0, None) == 0
    assert lib.lime_multi_copy(None, 33, None) == -1
    assert lib.lime_gather_rows_multi(None, 4, None, 2, None) == -1
    assert lib.lime_gather_rows_multi(None, 0, None, 2, None) == 0
    # the bf16 encoder-block entry points: NULL / unsupported shapes are refused before any launch
    assert lib.lime_encoder_ffn_bf16(None, None) == -1 and lib.lime_encoder_block_bf16(None, None) == -1 and lib.lime_inproj_bf16(None, None) == -1
    assert lib.lime_encoder_ffn_bf16(ctypes.byref(_lib.FfnBf16Args()), None) == -1
    assert lib.lime_ffn_pack_bf16(None, 300, None, None, 512, 300, 512, None, None, None) == -1
    assert lib.lime_ffn_bf16_model_columns() == 304
    assert lib.lime_ffn_pack_bf16_size(512, 0) == 512 * 320 and lib.lime_ffn_pack_bf16_size(512, 1) == 512 * 304
    assert lib.lime_oproj_pack_bf16_size() == 10 * 304 * 32 and lib.lime_inproj_pack_bf16_size(960) == 960 * 320
    assert lib.lime_sage_mean_f32(None, None, None, 1, 1, 1, 1, 1, None) == -1
    with pytest.raises(_lib.LimeHipError):
        _lib.check(-1, 'x')


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.LimeHipError):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from lime_cikm25_amd import ops
    with pytest.raises(TypeError):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))
