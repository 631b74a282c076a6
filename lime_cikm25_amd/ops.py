"""Tensor-level wrappers over the C ABI (include/lime_hip.h): shape/dtype/device checks on the host,
then one call into liblime_hip.so on torch's current stream.  PyTorch is used here for device memory
and streams only -- every wrapper launches hand-written HIP kernels and nothing else.

2-D operands may be row-strided views (``stride(1) == 1``); the row stride is passed as the leading
dimension, so slices like ``qkv[:, 300:600]`` or ``fused[:, 900:]`` cost nothing.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import LinearArgs, LIME_ACT, check


# bench.py sets this to a list to collect (variant, M, N, K, start_event, end_event) per lime_linear_f32 launch:
# HIP events recorded on the launch stream right around the launch (never used otherwise)
PROFILE = None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _mat(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError('%s must be a CUDA tensor (the HIP path has no CPU fallback)' % name)
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        raise ValueError('%s must be 2-D with unit column stride, got shape %s strides %s' % (name, tuple(t.shape), t.stride()))
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _vec(t, name, n=None, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise TypeError('%s must be a contiguous CUDA %s tensor' % (name, dtype))
    if n is not None and t.numel() != n:
        raise ValueError('%s must have %d elements, got %d' % (name, n, t.numel()))
    return t


def _mask_u8(mask, name):
    """bool/uint8 mask -> uint8 view (zero-copy)."""
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        mask = mask.contiguous().view(torch.uint8)
    return _vec(mask, name, dtype=torch.uint8)


def set_split_gemm(on, force=False):
    """Route the big lime_linear_f32 problems through the split-product kernel (True, the default: fp32-level products on the bf16
    matrix cores, csrc/gemm_sp_f32.hip) or the fp32-MFMA kernels (False).  ``force``: also the launches the dispatcher leaves to the
    other kernels because few 256-row tiles would fill the chip badly, and -- with ``on`` False -- the big-M fp32 kernel for problems the
    dispatcher gives to the 64-row-tile kernel for the same reason (tests).  Returns the previous setting (True / False)."""
    return bool(_lib.load().lime_set_split_gemm((5 if force else 1) if on else (4 if force else 0)) & 1)


def linear(a, w, bias=None, act=None, out=None, a_ids=None, a_pe=None, a_period=0, res=None, res_div=1, res_ids=None,
           res_pe=None, res_period=0, ln=None, ln_eps=1e-5, res_mod=0, pool32=False, ln_rstd=None, n_alg=None, m_dev=None,
           c_ids=None, act_scale=1.0, dropout=None, _build_only=False):
    """C = epilogue(A . W^T + bias) -- see ``lime_linear_f32`` in include/lime_hip.h.

    n_alg: the number of USEFUL output columns when w carries zero padding rows (in_proj with heads padded to 32 columns:
    900 of 960); only bench.py's per-kernel FLOP accounting reads it.
    m_dev: int32 device tensor (1 element): the launch computes min(m_dev, M) rows, M being the capacity of the buffers.
    c_ids: int32 [M]: result row r is stored at out[c_ids[r]] (out is then passed in, any number of rows) and the periodic
    residual is res[c_ids[r] % res_mod] -- the in_proj over a compacted token list (``compact_sequences``).

    a: [M, K] (or the [V, K] table when a_ids is given, M = len(a_ids)); w: [N, K]; out: [M, N] (may be a view).
    ln: (gamma, beta) for the fused LayerNorm (N <= 320); ln_rstd: optional [M] output of the rows' 1 / sqrt(var + eps).
    """
    lib = _lib.load()
    _mat(a, 'a')
    _mat(w, 'w')
    N, K = w.shape
    if a.shape[1] != K:
        raise ValueError('a has %d columns, w has K=%d' % (a.shape[1], K))
    if a_ids is not None:
        _vec(a_ids, 'a_ids', dtype=torch.int32)
        M = a_ids.numel()
    else:
        M = a.shape[0]
    rows_out = M // 32 if pool32 else M
    if pool32 and M % 32:
        raise ValueError('pool32 needs M %% 32 == 0 (M = %d)' % M)
    if out is None:
        out = torch.empty((rows_out, N), dtype=torch.float32, device=a.device)
    _mat(out, 'out')
    if c_ids is not None:
        if out.shape[1] != N or res_mod <= 0:
            raise ValueError('c_ids needs an [*, N] out and a periodic residual (res_mod > 0)')
        args_c_ids = _vec(c_ids, 'c_ids', M, dtype=torch.int32)
    elif tuple(out.shape) != (rows_out, N):
        raise ValueError('out must be [%d, %d], got %s' % (rows_out, N, tuple(out.shape)))
    args = LinearArgs()
    if c_ids is not None:
        args.c_ids = args_c_ids.data_ptr()
    if m_dev is not None:
        args.m_dev = _vec(m_dev, 'm_dev', 1, dtype=torch.int32).data_ptr()
    args.pool32 = 1 if pool32 else 0
    args.a, args.lda = a.data_ptr(), _ld(a)
    args.a_ids = a_ids.data_ptr() if a_ids is not None else None
    if a_pe is not None:
        _mat(a_pe, 'a_pe')
        if a_pe.shape[1] != K or a_pe.shape[0] < a_period or a_period <= 0:
            raise ValueError('a_pe must be [>=a_period, K]')
        args.a_pe, args.lda_pe, args.a_period = a_pe.data_ptr(), _ld(a_pe), a_period
    args.w, args.ldw = w.data_ptr(), _ld(w)
    args.bias = _vec(bias, 'bias', N).data_ptr() if bias is not None else None
    if res is not None:
        _mat(res, 'res')
        if res.shape[1] != N:
            raise ValueError('res must have N=%d columns' % N)
        args.res, args.ldr, args.res_div = res.data_ptr(), _ld(res), res_div
        if res_ids is not None:
            _vec(res_ids, 'res_ids', M, dtype=torch.int32)
            args.res_ids = res_ids.data_ptr()
            if res_pe is not None:
                _mat(res_pe, 'res_pe')
                if res_pe.shape[1] != N or res_pe.shape[0] < res_period or res_period <= 0:
                    raise ValueError('res_pe must be [>=res_period, N]')
                args.res_pe, args.ldr_pe, args.res_period = res_pe.data_ptr(), _ld(res_pe), res_period
        elif res_mod > 0:
            if res.shape[0] < res_mod:
                raise ValueError('res has %d rows, res_mod is %d' % (res.shape[0], res_mod))
            args.res_mod = res_mod
        elif res.shape[0] * res_div < M:
            raise ValueError('res has %d rows, needs >= %d' % (res.shape[0], (M + res_div - 1) // res_div))
    if ln is not None:
        args.ln_gamma = _vec(ln[0], 'ln gamma', N).data_ptr()
        args.ln_beta = _vec(ln[1], 'ln beta', N).data_ptr()
        args.ln_eps = ln_eps
        if ln_rstd is not None:
            args.ln_rstd = _vec(ln_rstd, 'ln_rstd', M).data_ptr()
    args.c, args.ldc = out.data_ptr(), _ld(out)
    args.M, args.N, args.K = M, N, K
    args.act = LIME_ACT[act]
    args.act_scale = act_scale
    if dropout is not None and dropout[0] > 0:          # (p, seed, site): nn.Dropout behind the activation, counter-based mask
        args.dropout_p, args.dropout_seed, args.dropout_site = dropout
    if ln is not None and M >= 4096 and not _SLOW_LN_WARNED and not _friendly16(a, w, out, res):
        # the LayerNorm epilogue of a big problem whose operands are not 16-byte friendly runs on the general kernel's 5-tile-wide
        # instantiation (256 registers in scratch, a tenth of the LDS-DMA kernels' rate): say so once instead of being silently slow
        import warnings
        warnings.warn('lime_linear_f32: LayerNorm epilogue on operands that are not 16-byte aligned with leading dimensions that are '
                      'multiples of 4 (M = %d): this takes the slow general kernel; pass contiguous / aligned views' % M, RuntimeWarning)
        globals()['_SLOW_LN_WARNED'] = True
    if _build_only:
        return args, out
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.lime_linear_f32(ctypes.byref(args), _stream()), 'lime_linear_f32')
        e1.record()
        m_run = M if m_dev is None else min(M, int(m_dev.item()))      # the rows this launch computed
        gathered = m_run * K * 4 if a_ids is not None else 0          # bytes of table rows this launch gathered as its A operand
        PROFILE.append((lib.lime_last_linear_kernel().decode(), m_run, N, K, n_alg or N, e0, e1, gathered))
        return out
    check(lib.lime_linear_f32(ctypes.byref(args), _stream()), 'lime_linear_f32')
    return out


_SLOW_LN_WARNED = False


def _friendly16(*tensors):
    return all(t is None or (t.data_ptr() % 16 == 0 and _ld(t) % 4 == 0) for t in tensors)


GROUP_SMALL_GEMMS = True      # False: linear_group issues its problems one by one (A/B runs, tests)


def linear_group(problems):
    """Several INDEPENDENT small GEMMs in one launch (``lime_linear_group_f32``): ``problems`` is a list of dicts of ``linear``'s
    arguments; returns the list of outputs.  The launches around the encoders are latency bound (>= 10 us each however small), so two
    that do not depend on each other cost one launch's time side by side.  Falls back to separate launches when a problem is outside
    the mid-M kernel (large M, LayerNorm, misaligned operands), under bench.py's per-kernel timing pass, or with GROUP_SMALL_GEMMS off."""
    if len(problems) == 1 or not GROUP_SMALL_GEMMS or PROFILE is not None or len(problems) > 8:
        return [linear(**kw) for kw in problems]
    lib = _lib.load()
    built = [linear(_build_only=True, **kw) for kw in problems]
    # only problems lime_linear_f32 itself hands to the mid-M kernel: M < 4096, or below 12288 rows with an epilogue the big-M kernels
    # take from there on only (tanh / sigmoid, broadcast or gathered residual without LayerNorm)
    mid = lambda a: a.M < 4096 or (a.M < 12288 and (a.act in (2, 3) or (a.res and (a.res_ids or a.res_div > 1))))
    if any(not mid(a) or a.ln_gamma or a.pool32 or a.c_ids or a.a_pe or a.res_pe or a.K % 4 or a.N % 4 or a.K < 16 for a, _ in built):
        return [linear(**kw) for kw in problems]
    arr = (LinearArgs * len(built))()
    for i, (a, _) in enumerate(built):
        ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(a), ctypes.sizeof(LinearArgs))
    st = lib.lime_linear_group_f32(arr, len(built), _stream())
    if st == -2:                                       # LIME_ERR_UNSUPPORTED                # e.g. a misaligned view: the general kernels take it
        return [linear(**kw) for kw in problems]
    check(st, 'lime_linear_group_f32')
    return [out for _, out in built]


def embed_pe(ids, table, pe=None, period=0, out=None):
    """out[r] = table[ids[r]] + pe[r % period]; ids int32 [rows]."""
    lib = _lib.load()
    _vec(ids, 'ids', dtype=torch.int32)
    _mat(table, 'table')
    rows, dim = ids.numel(), table.shape[1]
    if out is None:
        out = torch.empty((rows, dim), dtype=torch.float32, device=table.device)
    _mat(out, 'out')
    if pe is not None:
        _mat(pe, 'pe')
    check(lib.lime_embed_pe_f32(_p(ids), _p(table), _ld(table), _p(pe), _ld(pe) if pe is not None else 0, period, _p(out),
                                _ld(out), rows, dim, _stream()), 'lime_embed_pe_f32')
    return out


def token_attention(q, k, v, n_seq, S, n_head, head_dim, scale, key_mask=None, out=None, head_stride=None, n_seq_dev=None, lse=None):
    """softmax(Q K^T * scale [+mask]) V per (sequence, head); q/k/v: [n_seq*S, n_head*head_stride] views of one pitch
    (head_stride defaults to head_dim, the packed layout); out: [n_seq*S, n_head*head_dim].  ``lse`` (float32 [n_seq*S * n_head], no
    mask): also filled with the softmax statistics ``token_attention_bwd(..., lse=lse)`` reads instead of recomputing them."""
    lib = _lib.load()
    hs = head_dim if head_stride is None else head_stride
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v')):
        _mat(t, n)
        if tuple(t.shape) != (n_seq * S, n_head * hs):
            raise ValueError('%s must be [%d, %d]' % (n, n_seq * S, n_head * hs))
    if not (_ld(q) == _ld(k) == _ld(v)):
        raise ValueError('q, k and v must share one leading dimension')
    if out is None:
        out = torch.empty((n_seq * S, n_head * head_dim), dtype=torch.float32, device=q.device)
    _mat(out, 'out')
    m = _mask_u8(key_mask, 'key_mask')
    if m is not None and m.numel() != n_seq * S:
        raise ValueError('key_mask must have n_seq * S elements')
    if lse is not None:
        if m is not None or n_seq_dev is not None:
            raise ValueError('lse goes with the unmasked, host-counted call')
        _vec(lse, 'lse', n_seq * S * n_head)
        check(lib.lime_token_attention_lse_f32(_p(q), _p(k), _p(v), _ld(q), _p(out), _ld(out), _p(lse), n_seq, S, n_head, head_dim, hs,
                                               scale, _stream()), 'lime_token_attention_lse_f32')
        return out
    if n_seq_dev is not None:                      # a compacted batch: the sequence count lives on the device
        _vec(n_seq_dev, 'n_seq_dev', 1, dtype=torch.int32)
        check(lib.lime_token_attention_count_f32(_p(q), _p(k), _p(v), _ld(q), _p(m), _p(n_seq_dev), _p(out), _ld(out), n_seq, S, n_head,
                                                 head_dim, hs, scale, _stream()), 'lime_token_attention_count_f32')
        return out
    check(lib.lime_token_attention_f32(_p(q), _p(k), _p(v), _ld(q), _p(m), _p(out), _ld(out), n_seq, S, n_head, head_dim, hs,
                                       scale, _stream()), 'lime_token_attention_f32')
    return out


def token_attention_rows(q, k, v, row_map, n_seq_dev, n_seq, S, n_head, head_dim, scale, out=None):
    """``lime_token_attention_rows_f32``: attention over compacted sequences.  q / k / v: column views (heads 32 columns apart) of
    one buffer that also holds the S rows shared by the padding tokens; row_map: int32 [n_seq * S] -> row of that buffer;
    n_seq_dev: int32 device tensor (1 element) or None; out: [n_seq * S, n_head * head_dim] (rows of sequences beyond the
    device count are left untouched)."""
    lib = _lib.load()
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v')):
        _mat(t, n)
        if t.shape[1] != n_head * 32:
            raise ValueError('%s must have n_head * 32 columns' % n)
    if not (_ld(q) == _ld(k) == _ld(v)):
        raise ValueError('q, k and v must share one leading dimension')
    _vec(row_map, 'row_map', dtype=torch.int32)
    if row_map.numel() < n_seq * S:
        raise ValueError('row_map must cover n_seq * S tokens')
    if n_seq_dev is not None:
        _vec(n_seq_dev, 'n_seq_dev', 1, dtype=torch.int32)
    if out is None:
        out = torch.empty((n_seq * S, n_head * head_dim), dtype=torch.float32, device=q.device)
    _mat(out, 'out')
    check(lib.lime_token_attention_rows_f32(_p(q), _p(k), _p(v), _ld(q), _p(row_map), _p(n_seq_dev), _p(out), _ld(out), n_seq, S,
                                            n_head, head_dim, scale, _stream()), 'lime_token_attention_rows_f32')
    return out


class Compacted:
    """Index lists of ``compact_sequences`` (all int32 device tensors; see lime_compact_sequences in include/lime_hip.h)."""
    __slots__ = ('n_seq', 'S', 'cap', 'seq_inv', 'seq_src', 'ids_c', 'row_map', 'tok_ids', 'tok_rows', 'counts')

    n_compact = property(lambda self: self.counts[0:1])      # device counts as 1-element views (m_dev / n_seq_dev arguments)
    n_rows = property(lambda self: self.counts[1:2])
    n_live_tokens = property(lambda self: self.counts[2:3])
    n_tokens_and_pad_rows = property(lambda self: self.counts[4:5])     # live tokens + the S padding-row entries behind them in tok_ids / tok_rows


def compact_sequences(ids, pad_base=None):
    """ids int32 [n_seq, S] -> ``Compacted``: the live sequences (+ one all-padding representative) and their live tokens.
    pad_base: the q/k/v row where the S padding rows start (default: right behind the (n_seq + 1) * S compact rows)."""
    lib = _lib.load()
    if ids.dim() != 2 or ids.dtype != torch.int32 or not ids.is_cuda or not ids.is_contiguous():
        raise TypeError('ids must be a contiguous CUDA int32 [n_seq, S] tensor')
    n_seq, S = ids.shape
    cap = (n_seq + 1) * S
    c = Compacted()
    c.n_seq, c.S, c.cap = n_seq, S, cap
    dev = ids.device
    buf = torch.empty(n_seq + 4 * cap + 8 + int(lib.lime_compact_sequences_workspace(n_seq)), dtype=torch.int32, device=dev)
    c.seq_inv = buf[:n_seq]
    o = n_seq
    c.ids_c, c.row_map, c.tok_ids, c.tok_rows = (buf[o + i * cap:o + (i + 1) * cap] for i in range(4))
    o += 4 * cap
    c.counts = buf[o:o + 5]
    work = buf[o + 8:]
    c.seq_src = work[n_seq:2 * n_seq + 1]          # compact -> original sequence (-1: the all-padding representative)
    check(lib.lime_compact_sequences(_p(ids), n_seq, S, cap if pad_base is None else pad_base, _p(c.seq_inv), _p(c.ids_c), _p(c.row_map),
                                     _p(c.tok_ids), _p(c.tok_rows), _p(c.counts), _p(work), _stream()), 'lime_compact_sequences')
    return c


def pad_heads(src, n_blk, head_dim, head_stride):
    """[n_blk * head_dim, cols] (or a vector of n_blk * head_dim) -> rows padded with zeros to head_stride per block."""
    lib = _lib.load()
    vec = src.dim() == 1
    s2 = src.view(-1, 1) if vec else src
    _mat(s2, 'src')
    if s2.shape[0] != n_blk * head_dim:
        raise ValueError('src must have n_blk * head_dim rows')
    dst = torch.empty((n_blk * head_stride, s2.shape[1]), dtype=torch.float32, device=src.device)
    check(lib.lime_pad_heads_f32(_p(s2), _ld(s2), _p(dst), _ld(dst), n_blk, head_dim, head_stride, s2.shape[1], _stream()),
          'lime_pad_heads_f32')
    return dst.view(-1) if vec else dst


def mean_pool(x, n_seq, S, out=None, n_seq_dev=None):
    """out[s] = mean of the S rows of sequence s.  n_seq_dev: optional device-side int32 count (a compacted batch): only the
    first min(n_seq, count) sequences are read and written."""
    lib = _lib.load()
    _mat(x, 'x')
    if x.shape[0] != n_seq * S:
        raise ValueError('x must have n_seq * S rows')
    dim = x.shape[1]
    if out is None:
        out = torch.empty((n_seq, dim), dtype=torch.float32, device=x.device)
    _mat(out, 'out')
    if n_seq_dev is not None:
        _vec(n_seq_dev, 'n_seq_dev', dtype=torch.int32)
        check(lib.lime_mean_pool_count_f32(_p(x), _ld(x), _p(out), _ld(out), n_seq, S, dim, _p(n_seq_dev), _stream()),
              'lime_mean_pool_count_f32')
    else:
        check(lib.lime_mean_pool_f32(_p(x), _ld(x), _p(out), _ld(out), n_seq, S, dim, _stream()), 'lime_mean_pool_f32')
    return out


def bucketize(x, cuts=None):
    """int32 lifetime buckets (newsEncoders.py:53-58), bit-exact by threshold comparison.  cuts: fp32 device tensor of the
    num_buckets - 1 ascending cut points for num_buckets != 10 (the default table is built into the kernel)."""
    lib = _lib.load()
    x = _vec(x.contiguous(), 'x')
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    if cuts is not None:
        _vec(cuts, 'cuts')
        check(lib.lime_bucketize_cuts_f32(_p(x), _p(cuts), cuts.numel(), _p(out), x.numel(), _stream()), 'lime_bucketize_cuts_f32')
        return out
    check(lib.lime_bucketize_f32(_p(x), _p(out), x.numel(), _stream()), 'lime_bucketize_f32')
    return out


def mhsa_live_ids(ids, mask):
    """``lime_mhsa_live_ids``: ids int32 [n, T], mask bool/uint8 [n, T] -> ids with the -1 sentinel of all-zero sequences that must be encoded."""
    lib = _lib.load()
    n, T = ids.shape
    m = _mask_u8(mask, 'mask')
    out = torch.empty_like(ids)
    check(lib.lime_mhsa_live_ids(_p(ids), _p(m), n, T, _p(out), _stream()), 'lime_mhsa_live_ids')
    return out


def mhsa_compact_mask(cmp, mask):
    """``lime_mhsa_compact_mask``: clamps ``cmp.ids_c`` in place and returns the key mask in compact order, uint8 [(n_seq + 1), S]."""
    lib = _lib.load()
    m = _mask_u8(mask, 'mask')
    mask_c = torch.empty((cmp.n_seq + 1, cmp.S), dtype=torch.uint8, device=cmp.ids_c.device)
    check(lib.lime_mhsa_compact_mask(_p(cmp.ids_c), _p(cmp.seq_src), _p(m), cmp.n_seq + 1, cmp.S, _p(mask_c), _stream()), 'lime_mhsa_compact_mask')
    return mask_c


def fuse_rows(a, b, gate=None, out=None):
    """LIME's 'add' / 'gated' fusion (newsEncoders.py:154-159): a + b, or gate * a + (1 - gate) * b; [rows, cols] matrices."""
    lib = _lib.load()
    _mat(a, 'a')
    _mat(b, 'b')
    if a.shape != b.shape or (gate is not None and gate.shape != a.shape):
        raise ValueError('a, b (and gate) must have one shape')
    if gate is not None:
        _mat(gate, 'gate')
    if out is None:
        out = torch.empty(tuple(a.shape), dtype=torch.float32, device=a.device)
    _mat(out, 'out')
    check(lib.lime_fuse_rows_f32(_p(a), _ld(a), _p(b), _ld(b), _p(gate), _ld(gate) if gate is not None else 0, _p(out), _ld(out),
                                 a.shape[0], a.shape[1], _stream()), 'lime_fuse_rows_f32')
    return out


def gather_rows(idx, table, out):
    lib = _lib.load()
    _vec(idx, 'idx', dtype=torch.int32)
    _mat(table, 'table')
    _mat(out, 'out')
    if out.shape[0] != idx.numel() or out.shape[1] != table.shape[1]:
        raise ValueError('out must be [len(idx), table.shape[1]]')
    check(lib.lime_gather_rows_f32(_p(idx), _p(table), _ld(table), _p(out), _ld(out), idx.numel(), table.shape[1], _stream()),
          'lime_gather_rows_f32')
    return out


def topic_rep(cat, sub, cat_table, sub_table, w=None, bias=None, out=None, emb_out=None):
    """category_affine(cat[cat_emb, sub_emb]) into ``out`` and/or the raw embedding pair into ``emb_out``."""
    lib = _lib.load()
    _vec(cat, 'cat', dtype=torch.int32)
    _vec(sub, 'sub', cat.numel(), dtype=torch.int32)
    _mat(cat_table, 'cat_table')
    _mat(sub_table, 'sub_table')
    if not (cat_table.is_contiguous() and sub_table.is_contiguous()):
        raise ValueError('embedding tables must be contiguous')
    rows, dc, ds = cat.numel(), cat_table.shape[1], sub_table.shape[1]
    dout = 0
    if w is not None:
        _mat(w, 'w')
        if not w.is_contiguous() or w.shape[1] != dc + ds:
            raise ValueError('w must be contiguous [dout, dc + ds]')
        dout = w.shape[0]
        if out is None:
            out = torch.empty((rows, dout), dtype=torch.float32, device=cat.device)
        _mat(out, 'out')
        if tuple(out.shape) != (rows, dout):
            raise ValueError('out must be [rows, dout]')
        _vec(bias, 'bias', dout)
    if emb_out is not None:
        _mat(emb_out, 'emb_out')
        if tuple(emb_out.shape) != (rows, dc + ds):
            raise ValueError('emb_out must be [rows, dc + ds]')
    check(lib.lime_topic_rep_f32(_p(cat), _p(sub), _p(cat_table), _p(sub_table), dc, ds, _p(w), _p(bias), dout,
                                 _p(out) if w is not None else None, _ld(out) if w is not None else 0, _p(emb_out),
                                 _ld(emb_out) if emb_out is not None else 0, rows, _stream()), 'lime_topic_rep_f32')
    return out


def intent_fuse(intents, att_hidden, affine2_t, affine2_b, content, M, k, D, A):
    """intents [2*M*k, D], att_hidden [2*M*k, A] contiguous; writes content[:, :2D] (content may be a view)."""
    lib = _lib.load()
    _mat(intents, 'intents')
    _mat(att_hidden, 'att_hidden')
    if not (intents.is_contiguous() and att_hidden.is_contiguous()):
        raise ValueError('intents / att_hidden must be contiguous')
    if tuple(intents.shape) != (2 * M * k, D) or tuple(att_hidden.shape) != (2 * M * k, A):
        raise ValueError('intents / att_hidden have the wrong shape')
    _mat(content, 'content')
    if content.shape[0] != M or content.shape[1] < 2 * D:
        raise ValueError('content must be [M, >= 2D]')
    check(lib.lime_intent_fuse_f32(_p(intents), _p(att_hidden), _p(_vec(affine2_t, 'affine2_t', A)),
                                   _p(_vec(affine2_b, 'affine2_b', A)), _p(content), _ld(content), M, k, D, A, _stream()),
          'lime_intent_fuse_f32')
    return content


def additive_pool(hidden, affine2, x, n_seq, S, mask=None, out=None, n_seq_dev=None):
    """layers.Attention over the S tokens of each sequence (layers.py:285-300).  ``n_seq_dev`` (int32 device tensor, 1 element): only
    that many sequences are pooled (a compacted batch), the other output rows are left alone."""
    lib = _lib.load()
    _mat(hidden, 'hidden')
    _mat(x, 'x')
    A, D = hidden.shape[1], x.shape[1]
    if hidden.shape[0] != n_seq * S or x.shape[0] != n_seq * S:
        raise ValueError('hidden and x must have n_seq * S rows')
    if out is None:
        out = torch.empty((n_seq, D), dtype=torch.float32, device=x.device)
    _mat(out, 'out')
    m = _mask_u8(mask, 'mask')
    if n_seq_dev is not None:
        _vec(n_seq_dev, 'n_seq_dev', 1, dtype=torch.int32)
    check(lib.lime_additive_pool_count_f32(_p(hidden), _ld(hidden), _p(_vec(affine2, 'affine2', A)), A, _p(x), _ld(x), D, _p(m),
                                           _p(n_seq_dev), _p(out), _ld(out), n_seq, S, _stream()), 'lime_additive_pool_f32')
    return out


def fill_pad_rows(ids, src, dst, S):
    """dst[r] = src[r % S] for the rows r whose id is the padding word 0 (``lime_fill_pad_rows_f32``); returns dst."""
    lib = _lib.load()
    _vec(ids, 'ids', dtype=torch.int32)
    _mat(src, 'src')
    _mat(dst, 'dst')
    if dst.shape[0] != ids.numel() or src.shape[0] < S or src.shape[1] != dst.shape[1]:
        raise ValueError('fill_pad_rows: dst must be [len(ids), cols], src [>= S, cols]')
    check(lib.lime_fill_pad_rows_f32(_p(ids), _p(src), _ld(src), _p(dst), _ld(dst), ids.numel(), S, dst.shape[1], _stream()),
          'lime_fill_pad_rows_f32')
    return dst


def additive_pool_bwd(hidden, affine2, x, dout, n_seq, S, mask=None):
    """Backward of ``additive_pool``: -> (dhidden [n_seq * S, A], daffine2 [A], dx [n_seq * S, D])."""
    lib = _lib.load()
    _mat(hidden, 'hidden')
    _mat(x, 'x')
    _mat(dout, 'dout')
    A, D = hidden.shape[1], x.shape[1]
    if hidden.shape[0] != n_seq * S or x.shape[0] != n_seq * S or tuple(dout.shape) != (n_seq, D):
        raise ValueError('hidden and x must have n_seq * S rows, dout must be [n_seq, D]')
    dh = torch.empty((n_seq * S, A), dtype=torch.float32, device=x.device)
    dx = torch.empty((n_seq * S, D), dtype=torch.float32, device=x.device)
    part = torch.empty((max(n_seq, 1), A), dtype=torch.float32, device=x.device)
    m = _mask_u8(mask, 'mask')
    check(lib.lime_additive_pool_bwd_f32(_p(hidden), _ld(hidden), _p(_vec(affine2, 'affine2', A)), A, _p(x), _ld(x), D, _p(m), _p(dout),
                                         _ld(dout), _p(dh), _ld(dh), _p(dx), _ld(dx), _p(part), n_seq, S, _stream()), 'lime_additive_pool_bwd_f32')
    da2 = colsum(part[:n_seq]) if n_seq else torch.zeros(A, dtype=torch.float32, device=x.device)
    return dh, da2, dx


CAND_ATTN_BY_HEAD = True      # False: the one-workgroup-per-row kernel (tests compare the two)


def cand_attn_weights(qp, kp, mask, B, N, H, D, n_head, hist_div=1):
    """agg [B, H] of layers.py:66-81.  ``hist_div`` > 1: kp [B / hist_div, H, D] and mask [B / hist_div, H] hold one history for hist_div
    consecutive rows (the candidates of one impression)."""
    lib = _lib.load()
    if B % hist_div:
        raise ValueError('B must be a multiple of hist_div')
    Bh = B // hist_div
    _vec(qp, 'qp', B * N * D)
    _vec(kp, 'kp', Bh * H * D)
    m = _mask_u8(mask, 'mask')
    if m.numel() != Bh * H:
        raise ValueError('mask must be [B / hist_div, H]')
    agg = torch.empty((B, H), dtype=torch.float32, device=qp.device)
    by_head = (N + H) * (D // n_head + 1) * 4 <= 64 * 1024
    if hist_div > 1 and not (CAND_ATTN_BY_HEAD and by_head):
        kp, m, hist_div = kp.view(Bh, H * D).repeat_interleave(hist_div, dim=0).view(-1), m.view(Bh, H).repeat_interleave(hist_div, dim=0), 1
    if CAND_ATTN_BY_HEAD and by_head:
        # (row, head)-parallel: two short launches through a workspace (B = 32 rows alone leave 7 of 8 CUs idle)
        n_ws = int(lib.lime_cand_attn_weights_workspace(B, N, H, n_head))
        ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=qp.device)
        check(lib.lime_cand_attn_weights_shared_f32(_p(qp), _p(kp), _p(m), _p(agg), B, N, H, D, n_head, hist_div, _p(ws), n_ws, _stream()),
              'lime_cand_attn_weights_shared_f32')
        return agg
    check(lib.lime_cand_attn_weights_f32(_p(qp), _p(kp), _p(m), _p(agg), B, N, H, D, n_head, _stream()),
          'lime_cand_attn_weights_f32')
    return agg


def gate_ln_sage(y, x, scale, bias, gamma, beta, eps, groups, H, D, row_div, n_src, node_const=None):
    """``gate_ln`` over the H history rows of each of `groups` user rows + the GraphSAGE aggregate of the result in one launch:
    -> (refined [groups * H, D], mean [groups, D]).  x / y are [groups // row_div, H, D] (one history per row_div rows)."""
    lib = _lib.load()
    src_rows = (groups // row_div) * H * D
    x = _vec(x.contiguous(), 'x', src_rows)
    y = _vec(y.contiguous(), 'y', src_rows)
    scale = _vec(scale.contiguous(), 'scale', groups * H)
    out = torch.empty((groups * H, D), dtype=torch.float32, device=x.device)
    mean = torch.empty((groups, D), dtype=torch.float32, device=x.device)
    nc = _vec(node_const.contiguous(), 'node_const', D) if node_const is not None else None
    check(lib.lime_gate_ln_sage_f32(_p(y), _p(x), _p(scale), _p(_vec(bias, 'bias', D)), _p(_vec(gamma, 'gamma', D)), _p(_vec(beta, 'beta', D)),
                                    eps, _p(out), _p(nc) if nc is not None else None, _p(mean), groups, H, D, row_div, n_src, _stream()),
          'lime_gate_ln_sage_f32')
    return out, mean


def gate_ln(y, x, scale, bias, gamma, beta, eps=1e-5):
    """LayerNorm(g * (s x) + (1 - g) x), g = sigmoid(s y + bias): the gated residual of layers.py:84-89."""
    lib = _lib.load()
    D = x.shape[-1]
    x = _vec(x.contiguous(), 'x')
    y = _vec(y.contiguous(), 'y', x.numel())
    rows = x.numel() // D
    out = torch.empty_like(x)
    check(lib.lime_gate_ln_f32(_p(y), _p(x), _p(_vec(scale.contiguous(), 'scale', rows)), _p(_vec(bias, 'bias', D)),
                               _p(_vec(gamma, 'gamma', D)), _p(_vec(beta, 'beta', D)), eps, _p(out), rows, D, _stream()),
          'lime_gate_ln_f32')
    return out


def sage_mean(hist, user_nodes, B, H, n_src, D):
    lib = _lib.load()
    _vec(hist, 'hist', B * H * D)
    _vec(user_nodes, 'user_nodes')
    n_user = user_nodes.numel() // D
    out = torch.empty((B, D), dtype=torch.float32, device=hist.device)
    check(lib.lime_sage_mean_f32(_p(hist), _p(user_nodes), _p(out), B, H, n_user, n_src, D, _stream()), 'lime_sage_mean_f32')
    return out


def interest_match(kp, qp, g, cand, remaining, B, N, H, A, D, scale, alpha, beta, use_weight, use_penalty, want_logits=True,
                   want_user=True):
    lib = _lib.load()
    _vec(kp, 'kp', B * H * A)
    _vec(qp, 'qp', B * N * A)
    _vec(g, 'g', B * H * D)
    _vec(cand, 'cand', B * N * D)
    if want_logits and use_weight:
        remaining = _vec(remaining.contiguous(), 'remaining', B * N)
    user = torch.empty((B, N, D), dtype=torch.float32, device=kp.device) if want_user else None
    logits = torch.empty((B, N), dtype=torch.float32, device=kp.device) if want_logits else None
    check(lib.lime_interest_match_f32(_p(kp), _p(qp), _p(g), _p(cand), _p(remaining) if want_logits and use_weight else None,
                                      _p(user), _p(logits), B, N, H, A, D, scale, alpha, beta, int(use_weight),
                                      int(use_penalty), _stream()), 'lime_interest_match_f32')
    return user, logits


def lifetime_score(user, news, remaining, alpha, beta, use_weight, use_penalty):
    lib = _lib.load()
    D = user.shape[-1]
    user = _vec(user.contiguous(), 'user')
    news = _vec(news.contiguous(), 'news', user.numel())
    rows = user.numel() // D
    if use_weight:
        remaining = _vec(remaining.contiguous(), 'remaining', rows)
    logits = torch.empty(user.shape[:-1], dtype=torch.float32, device=user.device)
    check(lib.lime_lifetime_score_f32(_p(user), _p(news), _p(remaining) if use_weight else None, _p(logits), rows, D, alpha,
                                      beta, int(use_weight), int(use_penalty), _stream()), 'lime_lifetime_score_f32')
    return logits


def row_scale(x, scale):
    lib = _lib.load()
    D = x.shape[-1]
    x = _vec(x.contiguous(), 'x')
    rows = x.numel() // D
    scale = _vec(scale.contiguous(), 'scale', rows)
    out = torch.empty_like(x)
    check(lib.lime_row_scale_f32(_p(x), _p(scale), _p(out), rows, D, _stream()), 'lime_row_scale_f32')
    return out


def inv_sqrt(x):
    return 1.0 / math.sqrt(float(x))


def multi_copy(pairs):
    """dst.copy_(src) for every (dst, src) pair in ONE launch per 32 pairs (same dtype / shape, both contiguous CUDA)."""
    lib = _lib.load()
    todo = []
    for dst, src in pairs:
        if dst.shape != src.shape or dst.dtype != src.dtype or not dst.is_cuda or not src.is_cuda:
            raise ValueError('multi_copy: shape / dtype / device mismatch (%s %s vs %s %s)' % (tuple(dst.shape), dst.dtype,
                                                                                              tuple(src.shape), src.dtype))
        if not dst.is_contiguous():
            raise ValueError('multi_copy: destinations must be contiguous')
        if not src.is_contiguous():
            src = src.contiguous()
        if dst.data_ptr() != src.data_ptr() and dst.numel():
            todo.append((dst, src))
    for i in range(0, len(todo), _lib.MAX_COPIES):
        part = todo[i:i + _lib.MAX_COPIES]
        table = (_lib.CopyDesc * len(part))()
        for d, (dst, src) in zip(table, part):
            d.src, d.dst, d.bytes = src.data_ptr(), dst.data_ptr(), dst.numel() * dst.element_size()
        check(lib.lime_multi_copy(table, len(part), _stream()), 'lime_multi_copy')
    return todo            # keeps the sources referenced until the caller drops the list (the copy is asynchronous)


# ---- bf16 matrix-core path (BASELINE config 3) ---------------------------------------------------------------------------
def to_bf16(src, rows_out=None, cols_out=None, out=None):
    """fp32 [rows, cols] (or a vector) -> bf16 [rows_out, cols_out], zero padded; cols_out a multiple of 4."""
    lib = _lib.load()
    vec = src.dim() == 1
    s2 = src.view(-1, 1) if vec else src
    _mat(s2, 'src')
    rows, cols = s2.shape
    if vec:                                       # a vector is padded along its only axis
        s2 = src.view(1, -1)
        rows, cols = 1, src.numel()
        rows_out = 1
    rows_out = rows if rows_out is None else rows_out
    cols_out = (cols + 7) // 8 * 8 if cols_out is None else cols_out
    if out is None:
        out = torch.empty((rows_out, cols_out), dtype=torch.bfloat16, device=src.device)
    _mat(out, 'out', dtype=torch.bfloat16)
    check(lib.lime_to_bf16(_p(s2), _ld(s2), rows, cols, _p(out), _ld(out), rows_out, cols_out, _stream()), 'lime_to_bf16')
    return out.view(-1) if vec else out


def linear_bf16(a, w, bias=None, act=None, out=None, a_ids=None, res=None, res_kind=0, res_mod=0, res_ids=None, res_pe=None,
                res_period=0, ln=None, ln_eps=1e-5, ln_count=None, pool32=False, n_alg=None, k_alg=None, m_dev=None, c_ids=None):
    """``lime_linear_bf16``: a / w / out (and residual kinds 2, 3) bfloat16, bias / LayerNorm / residual kind 1 fp32."""
    lib = _lib.load()
    _mat(a, 'a', dtype=torch.bfloat16)
    _mat(w, 'w', dtype=torch.bfloat16)
    N, K = w.shape
    if a.shape[1] != K:
        raise ValueError('a has %d columns, w has K=%d' % (a.shape[1], K))
    M = a_ids.numel() if a_ids is not None else a.shape[0]
    if pool32 and M % 32:
        raise ValueError('pool32 needs M %% 32 == 0 (M = %d)' % M)
    rows_out, odt = (M // 32, torch.float32) if pool32 else (M, torch.bfloat16)       # pool32: fp32 means over 32-row blocks
    if out is None:
        out = torch.empty((rows_out, N), dtype=odt, device=a.device)
    _mat(out, 'out', dtype=odt)
    if c_ids is not None:
        if out.shape[1] != N:
            raise ValueError('out must have N columns')
    elif tuple(out.shape) != (rows_out, N):
        raise ValueError('out must be [%d, %d], got %s' % (rows_out, N, tuple(out.shape)))
    args = _lib.LinearBf16Args()
    if c_ids is not None:
        args.c_ids = _vec(c_ids, 'c_ids', M, dtype=torch.int32).data_ptr()
    if m_dev is not None:
        args.m_dev = _vec(m_dev, 'm_dev', 1, dtype=torch.int32).data_ptr()
    args.pool32 = 1 if pool32 else 0
    args.a, args.lda = a.data_ptr(), _ld(a)
    args.a_ids = _vec(a_ids, 'a_ids', dtype=torch.int32).data_ptr() if a_ids is not None else None
    args.w, args.ldw = w.data_ptr(), _ld(w)
    args.bias = _vec(bias, 'bias', N).data_ptr() if bias is not None else None
    if res is not None:
        if res_kind not in (1, 2, 3):
            raise ValueError('res_kind must be 1 (fp32 rows), 2 (gathered bf16 rows) or 3 (bf16 rows)')
        _mat(res, 'res', dtype=torch.float32 if res_kind == 1 else torch.bfloat16)
        if res.shape[1] != N:
            raise ValueError('res must have N=%d columns' % N)
        args.res, args.ldr, args.res_kind, args.res_mod = res.data_ptr(), _ld(res), res_kind, res_mod
        if res_kind == 1 and res.shape[0] < (res_mod if res_mod > 0 else M):
            raise ValueError('res has too few rows')
        if res_kind == 3 and res.shape[0] < M:
            raise ValueError('res has too few rows')
        if res_kind == 2:
            args.res_ids = _vec(res_ids, 'res_ids', M, dtype=torch.int32).data_ptr()
            if res_pe is not None:
                _mat(res_pe, 'res_pe')
                if res_pe.shape[1] != N or res_pe.shape[0] < res_period or res_period <= 0:
                    raise ValueError('res_pe must be [>=res_period, N]')
                args.res_pe, args.ldr_pe, args.res_period = res_pe.data_ptr(), _ld(res_pe), res_period
    if ln is not None:
        args.ln_gamma = _vec(ln[0], 'ln gamma', N).data_ptr()
        args.ln_beta = _vec(ln[1], 'ln beta', N).data_ptr()
        args.ln_eps, args.ln_count = ln_eps, (N if ln_count is None else ln_count)
    args.c, args.ldc = out.data_ptr(), _ld(out)
    args.M, args.N, args.K = M, N, K
    args.act = LIME_ACT[act]
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.lime_linear_bf16(ctypes.byref(args), _stream()), 'lime_linear_bf16')
        e1.record()
        m_run = M if m_dev is None else min(M, int(m_dev.item()))
        PROFILE.append((lib.lime_last_linear_kernel().decode(), m_run, N, k_alg or K, n_alg or N, e0, e1))
        return out
    check(lib.lime_linear_bf16(ctypes.byref(args), _stream()), 'lime_linear_bf16')
    return out


def ffn_model_columns():
    """The column count (304) ``encoder_ffn_bf16`` carries the model dimension in."""
    return int(_lib.load().lime_ffn_bf16_model_columns())


def ffn_pack_bf16(w1, b1, w2):
    """``lime_ffn_pack_bf16``: linear1 / linear2 weights (fp32 [F, E], [F], [E, F]) -> the bf16 operands of ``encoder_ffn_bf16``."""
    lib = _lib.load()
    _mat(w1, 'w1')
    _mat(w2, 'w2')
    F, E = w1.shape
    if tuple(w2.shape) != (E, F):
        raise ValueError('w2 must be [%d, %d]' % (E, F))
    _vec(b1, 'b1', F)
    w1p = torch.empty(int(lib.lime_ffn_pack_bf16_size(F, 0)), dtype=torch.bfloat16, device=w1.device)
    w2p = torch.empty(int(lib.lime_ffn_pack_bf16_size(F, 1)), dtype=torch.bfloat16, device=w1.device)
    check(lib.lime_ffn_pack_bf16(_p(w1), _ld(w1), _p(b1), _p(w2), _ld(w2), E, F, _p(w1p), _p(w2p), _stream()), 'lime_ffn_pack_bf16')
    return w1p, w2p


def encoder_ffn_bf16(x, w1p, w2p, b2, ln, ln_eps, E, pool32=False, m_dev=None, out=None):
    """``lime_encoder_ffn_bf16``: LayerNorm(x + W2 relu(W1 x + b1) + b2) in one launch; x bf16 [M, 304] (E real columns).
    pool32: fp32 [M / 32, 304] means over 32-row blocks, else bf16 [M, 304]."""
    lib = _lib.load()
    _mat(x, 'x', dtype=torch.bfloat16)
    _vec(w1p, 'w1p', dtype=torch.bfloat16)
    _vec(w2p, 'w2p', dtype=torch.bfloat16)
    M, DP = x.shape
    F = w2p.numel() // ffn_model_columns()
    if DP != ffn_model_columns() or w1p.numel() != int(lib.lime_ffn_pack_bf16_size(F, 0)) or w2p.numel() != int(lib.lime_ffn_pack_bf16_size(F, 1)):
        raise ValueError('x [M, %d] and the two buffers of ffn_pack_bf16 expected' % ffn_model_columns())
    if pool32 and M % 32:
        raise ValueError('pool32 needs M %% 32 == 0 (M = %d)' % M)
    rows_out, odt = (M // 32, torch.float32) if pool32 else (M, torch.bfloat16)
    if out is None:
        out = torch.empty((rows_out, DP), dtype=odt, device=x.device)
    _mat(out, 'out', dtype=odt)
    if tuple(out.shape) != (rows_out, DP):
        raise ValueError('out must be [%d, %d]' % (rows_out, DP))
    args = _lib.FfnBf16Args()
    args.x, args.ldx = x.data_ptr(), _ld(x)
    args.w1p, args.w2p = w1p.data_ptr(), w2p.data_ptr()
    args.b2 = _vec(b2, 'b2', E).data_ptr()
    args.ln_gamma = _vec(ln[0], 'ln gamma', E).data_ptr()
    args.ln_beta = _vec(ln[1], 'ln beta', E).data_ptr()
    args.ln_eps = ln_eps
    args.pool32 = 1 if pool32 else 0
    args.out, args.ldo = out.data_ptr(), _ld(out)
    args.M, args.E, args.F = M, E, F
    if m_dev is not None:
        args.m_dev = _vec(m_dev, 'm_dev', 1, dtype=torch.int32).data_ptr()
    if PROFILE is not None:                        # bench.py: as ONE GEMM of the two layers' FLOPs: 2 M E (2 F)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.lime_encoder_ffn_bf16(ctypes.byref(args), _stream()), 'lime_encoder_ffn_bf16')
        e1.record()
        m_run = M if m_dev is None else min(M, int(m_dev.item()))
        PROFILE.append(('ffn_bf16_kernel<%s, false>' % ('true' if pool32 else 'false'), m_run, 2 * F, E, 2 * F, e0, e1))
        return out
    check(lib.lime_encoder_ffn_bf16(ctypes.byref(args), _stream()), 'lime_encoder_ffn_bf16')
    return out


def inproj_pack_bf16(w, K):
    """``lime_inproj_pack_bf16``: in_proj weight (fp32 [N, E], heads padded to 32 columns: N % 320 == 0) -> the bf16 ring slots of
    ``inproj_bf16``; K: the column count of the bf16 operand rows (E rounded up to 8; the columns beyond E get zero weights)."""
    lib = _lib.load()
    _mat(w, 'w')
    N = w.shape[0]
    if w.shape[1] > K:
        raise ValueError('K must cover the weight columns')
    wp = torch.empty(int(lib.lime_inproj_pack_bf16_size(N)), dtype=torch.bfloat16, device=w.device)
    check(lib.lime_inproj_pack_bf16(_p(w), _ld(w), N, w.shape[1], _p(wp), _stream()), 'lime_inproj_pack_bf16')
    return wp


def inproj_bf16(a, wp, add_rows, N, out, a_ids=None, c_ids=None, m_dev=None):
    """``lime_inproj_bf16``: out[c_ids[r] or r] = bf16(a[a_ids[r] or r] . w^T + add_rows[(c_ids[r] or r) % period]); a bf16 [*, K]."""
    lib = _lib.load()
    _mat(a, 'a', dtype=torch.bfloat16)
    _mat(out, 'out', dtype=torch.bfloat16)
    _mat(add_rows, 'add_rows')
    K = a.shape[1]
    M = a_ids.numel() if a_ids is not None else a.shape[0]
    if out.shape[1] != N or add_rows.shape[1] < N or wp.numel() != int(lib.lime_inproj_pack_bf16_size(N)):
        raise ValueError('out must have N columns, add_rows >= N columns, wp must come from inproj_pack_bf16')
    args = _lib.InprojBf16Args()
    args.a, args.lda, args.a_rows = a.data_ptr(), _ld(a), a.shape[0]
    args.a_ids = _vec(a_ids, 'a_ids', dtype=torch.int32).data_ptr() if a_ids is not None else None
    args.wp = _vec(wp, 'wp', dtype=torch.bfloat16).data_ptr()
    args.add_rows, args.ld_add, args.add_period = add_rows.data_ptr(), _ld(add_rows), add_rows.shape[0]
    args.M, args.N, args.K = M, N, K
    args.c_ids = _vec(c_ids, 'c_ids', M, dtype=torch.int32).data_ptr() if c_ids is not None else None
    args.out, args.ldo, args.out_rows = out.data_ptr(), _ld(out), out.shape[0]
    if m_dev is not None:
        args.m_dev = _vec(m_dev, 'm_dev', 1, dtype=torch.int32).data_ptr()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.lime_inproj_bf16(ctypes.byref(args), _stream()), 'lime_inproj_bf16')
        e1.record()
        m_run = M if m_dev is None else min(M, int(m_dev.item()))
        PROFILE.append(('inproj_bf16_kernel', m_run, N, K // 8 * 8 - (4 if K == 304 else 0), N * 30 // 32, e0, e1))
        return out
    check(lib.lime_inproj_bf16(ctypes.byref(args), _stream()), 'lime_inproj_bf16')
    return out


def oproj_pack_bf16(w):
    """``lime_oproj_pack_bf16``: out_proj weight (fp32 [E, E]) -> the bf16 ring slots of ``encoder_block_bf16``."""
    lib = _lib.load()
    _mat(w, 'w')
    E = w.shape[0]
    if w.shape[1] != E:
        raise ValueError('w must be square')
    wp = torch.empty(int(lib.lime_oproj_pack_bf16_size()), dtype=torch.bfloat16, device=w.device)
    check(lib.lime_oproj_pack_bf16(_p(w), _ld(w), E, _p(wp), _stream()), 'lime_oproj_pack_bf16')
    return wp


def _aligned16(t):
    return t if t.data_ptr() % 16 == 0 else t.clone()


def encoder_block_bf16(attn, w0p, add_rows, ln1, ln1_eps, res, res_kind, w1p, w2p, b2, ln2, ln2_eps, E, res_ids=None, pool32=False, m_dev=None,
                       out=None):
    """``lime_encoder_block_bf16``: out_proj + residual + norm1 + linear1 + ReLU + linear2 + residual + norm2 (+ 32-row block means) in
    one launch.  attn bf16 [M, 304]; res_kind 2: res = bf16 table gathered by res_ids, 3: res = bf16 rows [M, 304]; add_rows fp32
    [period, >= E]: out_proj's bias (+ the positional rows), row r % period goes to token r."""
    lib = _lib.load()
    _mat(attn, 'attn', dtype=torch.bfloat16)
    _mat(res, 'res', dtype=torch.bfloat16)
    _mat(add_rows, 'add_rows')
    M, DP = attn.shape
    F = w2p.numel() // ffn_model_columns()
    if DP != ffn_model_columns() or res.shape[1] != DP:
        raise ValueError('attn and res must have %d columns' % ffn_model_columns())
    if add_rows.shape[1] < E:
        raise ValueError('add_rows must have >= E columns')
    if (w0p.numel() != int(lib.lime_oproj_pack_bf16_size()) or w1p.numel() != int(lib.lime_ffn_pack_bf16_size(F, 0)) or
            w2p.numel() != int(lib.lime_ffn_pack_bf16_size(F, 1))):
        raise ValueError('w0p / w1p / w2p must come from oproj_pack_bf16 / ffn_pack_bf16')
    if pool32 and M % 32:
        raise ValueError('pool32 needs M %% 32 == 0 (M = %d)' % M)
    rows_out, odt = (M // 32, torch.float32) if pool32 else (M, torch.bfloat16)
    if out is None:
        out = torch.empty((rows_out, DP), dtype=odt, device=attn.device)
    _mat(out, 'out', dtype=odt)
    if tuple(out.shape) != (rows_out, DP):
        raise ValueError('out must be [%d, %d]' % (rows_out, DP))
    keep = [_aligned16(_vec(ln1[0], 'ln1 gamma', E)), _aligned16(_vec(ln1[1], 'ln1 beta', E))]
    args = _lib.EncoderBlockBf16Args()
    args.attn, args.lda = attn.data_ptr(), _ld(attn)
    args.w0p = _vec(w0p, 'w0p', dtype=torch.bfloat16).data_ptr()
    args.add_rows, args.ld_add, args.add_period = add_rows.data_ptr(), _ld(add_rows), add_rows.shape[0]
    args.ln1_gamma, args.ln1_beta, args.ln1_eps = keep[0].data_ptr(), keep[1].data_ptr(), ln1_eps
    args.res_kind = res_kind
    args.res, args.ldr, args.res_rows = res.data_ptr(), _ld(res), res.shape[0]
    if res_kind == 2:
        args.res_ids = _vec(res_ids, 'res_ids', M, dtype=torch.int32).data_ptr()
    elif res_kind == 3:
        if res.shape[0] < M:
            raise ValueError('res has too few rows')
    else:
        raise ValueError('res_kind must be 2 or 3')
    args.pool32 = 1 if pool32 else 0
    args.w1p, args.w2p = _vec(w1p, 'w1p', dtype=torch.bfloat16).data_ptr(), _vec(w2p, 'w2p', dtype=torch.bfloat16).data_ptr()
    args.b2 = _vec(b2, 'b2', E).data_ptr()
    args.ln2_gamma, args.ln2_beta, args.ln2_eps = _vec(ln2[0], 'ln2 gamma', E).data_ptr(), _vec(ln2[1], 'ln2 beta', E).data_ptr(), ln2_eps
    args.M, args.E, args.F = M, E, F
    args.out, args.ldo = out.data_ptr(), _ld(out)
    if m_dev is not None:
        args.m_dev = _vec(m_dev, 'm_dev', 1, dtype=torch.int32).data_ptr()
    if PROFILE is not None:                        # bench.py: as ONE GEMM of the block's FLOPs: 2 M (E E + 2 E F) = 2 M E (E + 2 F)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.lime_encoder_block_bf16(ctypes.byref(args), _stream()), 'lime_encoder_block_bf16')
        e1.record()
        m_run = M if m_dev is None else min(M, int(m_dev.item()))
        PROFILE.append(('ffn_bf16_kernel<%s, true>' % ('true' if pool32 else 'false'), m_run, DP + 2 * F, E, E + 2 * F, e0, e1))
        return out
    check(lib.lime_encoder_block_bf16(ctypes.byref(args), _stream()), 'lime_encoder_block_bf16')
    return out


def mean_pool_bf16(x, n_seq, S, dim, out=None):
    lib = _lib.load()
    _mat(x, 'x', dtype=torch.bfloat16)
    if x.shape[0] != n_seq * S:
        raise ValueError('x must have n_seq * S rows')
    if out is None:
        out = torch.empty((n_seq, dim), dtype=torch.float32, device=x.device)
    _mat(out, 'out')
    check(lib.lime_mean_pool_bf16(_p(x), _ld(x), _p(out), _ld(out), n_seq, S, dim, _stream()), 'lime_mean_pool_bf16')
    return out


def token_attention_rows_bf16(q, k, v, row_map, n_seq_dev, n_seq, S, n_head, head_dim, scale, out_cols=None, out=None):
    """``lime_token_attention_rows_bf16``: the bf16 attention over compacted sequences (see ``token_attention_rows``)."""
    lib = _lib.load()
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v')):
        _mat(t, n, dtype=torch.bfloat16)
        if t.shape[1] != n_head * 32:
            raise ValueError('%s must have n_head * 32 columns' % n)
    if not (_ld(q) == _ld(k) == _ld(v)):
        raise ValueError('q, k and v must share one leading dimension')
    _vec(row_map, 'row_map', dtype=torch.int32)
    if row_map.numel() < n_seq * S:
        raise ValueError('row_map must cover n_seq * S tokens')
    if n_seq_dev is not None:
        _vec(n_seq_dev, 'n_seq_dev', 1, dtype=torch.int32)
    out_cols = n_head * head_dim if out_cols is None else out_cols
    if out is None:
        out = torch.empty((n_seq * S, out_cols), dtype=torch.bfloat16, device=q.device)
    _mat(out, 'out', dtype=torch.bfloat16)
    check(lib.lime_token_attention_rows_bf16(_p(q), _p(k), _p(v), _ld(q), _p(row_map), _p(n_seq_dev), _p(out), _ld(out), n_seq, S, n_head,
                                             head_dim, scale, out_cols, _stream()), 'lime_token_attention_rows_bf16')
    return out


def token_attention_bf16(q, k, v, n_seq, S, n_head, head_dim, scale, out_cols=None, out=None):
    """Unmasked encoder attention on bf16 storage: q / k / v [n_seq * S, n_head * 32] bf16 views of one qkv buffer."""
    lib = _lib.load()
    for t, name in ((q, 'q'), (k, 'k'), (v, 'v')):
        _mat(t, name, dtype=torch.bfloat16)
        if t.shape[0] != n_seq * S or t.shape[1] != n_head * 32:
            raise ValueError('%s must be [n_seq * S, n_head * 32]' % name)
    if not (_ld(q) == _ld(k) == _ld(v)):
        raise ValueError('q, k, v must share one leading dimension')
    out_cols = n_head * head_dim if out_cols is None else out_cols
    if out is None:
        out = torch.empty((n_seq * S, out_cols), dtype=torch.bfloat16, device=q.device)
    _mat(out, 'out', dtype=torch.bfloat16)
    check(lib.lime_token_attention_bf16(_p(q), _p(k), _p(v), _ld(q), _p(out), _ld(out), n_seq, S, n_head, head_dim, scale, out_cols,
                                        _stream()), 'lime_token_attention_bf16')
    return out


def gather_rows_multi_prepare(n_rows, pairs):
    """Validate (table, out) pairs once and build the descriptor tables: returns an opaque plan for gather_rows_multi_run.
    Tables / outputs: any dtype, row-major with contiguous rows ([n, ...] -> rows of prod(shape[1:]) elements); every
    out has n_rows rows.  The plan keeps the tensors alive."""
    todo = []
    for table, out in pairs:
        if table.dtype != out.dtype or not table.is_cuda or not out.is_cuda or table.shape[1:] != out.shape[1:] or out.shape[0] != n_rows:
            raise ValueError('gather_rows_multi: table %s %s vs out %s %s (rows %d)' % (tuple(table.shape), table.dtype,
                                                                                        tuple(out.shape), out.dtype, n_rows))
        if not table.is_contiguous() or not out.is_contiguous():
            raise ValueError('gather_rows_multi: tables and outputs must be contiguous')
        todo.append((table, out))
    tables = []
    for i in range(0, len(todo), _lib.MAX_GATHERS):
        part = todo[i:i + _lib.MAX_GATHERS]
        descs = (_lib.GatherDesc * len(part))()
        for d, (table, out) in zip(descs, part):
            rb = (table.numel() // max(1, table.shape[0])) * table.element_size()
            d.table, d.table_stride, d.out, d.out_stride, d.row_bytes = table.data_ptr(), rb, out.data_ptr(), rb, rb
        tables.append((descs, len(part)))
    return {'n_rows': n_rows, 'tables': tables, 'keep': todo}


def gather_rows_multi_run(idx, plan):
    lib = _lib.load()
    if idx.dtype != torch.int32 or not idx.is_cuda or not idx.is_contiguous() or idx.numel() != plan['n_rows']:
        raise ValueError('gather_rows_multi: idx must be a contiguous int32 CUDA vector of %d rows' % plan['n_rows'])
    stream = _stream()
    for descs, n in plan['tables']:
        check(lib.lime_gather_rows_multi(idx.data_ptr(), plan['n_rows'], descs, n, stream), 'lime_gather_rows_multi')


def gather_rows_multi(idx, pairs):
    """out[r] = table[idx[r]] for every (table, out) pair, one launch per 16 pairs."""
    _vec(idx, 'idx', dtype=torch.int32)
    gather_rows_multi_run(idx, gather_rows_multi_prepare(idx.numel(), pairs))


# ---- training step (include/lime_hip.h, "Training step") ------------------------------------------------------------------
_WS = {}


def _workspace(device, floats):
    """A per-device scratch buffer for the fixed-order reductions (grown on demand; launches on one stream are ordered, so
    consecutive kernels can share it)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(max(int(floats), 1 << 20), dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def linear_wgrad(dy, x, out=None, accumulate=False, bias_out=None, want_bias=False):
    """dW [N, K] (+)= dy^T x  (dy [M, N], x [M, K]); with ``want_bias`` / ``bias_out`` also db [N] (+)= column sums of dy,
    returned as (dW, db)."""
    lib = _lib.load()
    _mat(dy, 'dy')
    _mat(x, 'x')
    M, N = dy.shape
    K = x.shape[1]
    if x.shape[0] != M:
        raise ValueError('dy has %d rows, x has %d' % (M, x.shape[0]))
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs out')
        out = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    _mat(out, 'out')
    if tuple(out.shape) != (N, K):
        raise ValueError('out must be [%d, %d]' % (N, K))
    if bias_out is None and want_bias:
        if accumulate:
            raise ValueError('accumulate needs bias_out')
        bias_out = torch.empty(N, dtype=torch.float32, device=dy.device)
    if bias_out is not None:
        _vec(bias_out, 'bias_out', N)
    if M == 0:
        if not accumulate:
            out.zero_()
            if bias_out is not None:
                bias_out.zero_()
    else:
        need = lib.lime_linear_wgrad_workspace(M, N, K)
        ws = _workspace(dy.device, need)
        if PROFILE is not None:                    # bench.py: HIP events around the launch pair (wgrad + partial-sum reduction)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(lib.lime_linear_wgrad_f32(_p(dy), _ld(dy), _p(x), _ld(x), _p(out), _ld(out), _p(bias_out), M, N, K,
                                        1 if accumulate else 0, _p(ws), ws.numel(), _stream()), 'lime_linear_wgrad_f32')
        if PROFILE is not None:
            e1.record()
            PROFILE.append(('wgrad_kernel + reduce_partials_kernel (dW = dY^T X)', M, N, K, N, e0, e1))
    return out if bias_out is None else (out, bias_out)


def colsum(x, out=None, accumulate=False):
    """out [N] (+)= column sums of x [M, N]."""
    lib = _lib.load()
    _mat(x, 'x')
    M, N = x.shape
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs out')
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    _vec(out, 'out', N)
    if M == 0:
        return out if accumulate else out.zero_()
    ws = _workspace(x.device, lib.lime_colsum_workspace(M, N))
    check(lib.lime_colsum_f32(_p(x), _ld(x), M, N, _p(out), 1 if accumulate else 0, _p(ws), ws.numel(), _stream()), 'lime_colsum_f32')
    return out


def layernorm_bwd(dy, y, gamma, beta, rstd, dy_div=1, dy_scale=1.0, want_dzsum=True, dropout=None):
    """Backward of y = LayerNorm(z): returns (dz [M, E], dgamma, dbeta, dzsum or None).  dy: [ceil(M / dy_div), E].
    ``dropout`` = (p, seed, site): a fifth result, dropout(dz) under that site's mask, written in the same pass; dzsum is then the
    column sums of THAT (the bias gradient of the linear in front of the dropout)."""
    lib = _lib.load()
    _mat(dy, 'dy')
    _mat(y, 'y')
    M, E = y.shape
    if dy.shape[1] != E or dy.shape[0] * dy_div < M:
        raise ValueError('dy must be [>= %d, %d]' % ((M + dy_div - 1) // dy_div, E))
    _vec(gamma, 'gamma', E)
    _vec(beta, 'beta', E)
    _vec(rstd, 'rstd', M)
    dev = y.device
    dz = torch.empty((M, E), dtype=torch.float32, device=dev)
    dgamma = torch.empty(E, dtype=torch.float32, device=dev)
    dbeta = torch.empty(E, dtype=torch.float32, device=dev)
    dzsum = torch.empty(E, dtype=torch.float32, device=dev) if want_dzsum else None
    ws = _workspace(dev, lib.lime_layernorm_bwd_workspace(M, E))
    if dropout is not None:
        if E % 4 or _ld(dy) % 4 or _ld(y) % 4:           # the fused copy needs 16-byte rows: otherwise a dropout pass of its own
            res = layernorm_bwd(dy, y, gamma, beta, rstd, dy_div, dy_scale, False)
            dt = globals()['dropout'](res[0], *dropout)
            return res[:3] + (colsum(dt) if want_dzsum else None, dt)
        dt = torch.empty((M, E), dtype=torch.float32, device=dev)
        check(lib.lime_layernorm_bwd_dropout_f32(_p(dy), _ld(dy), dy_div, dy_scale, _p(y), _ld(y), _p(gamma), _p(beta), _p(rstd), _p(dz),
                                                 _ld(dz), M, E, _p(dgamma), _p(dbeta), _p(dzsum), 0, _p(ws), ws.numel(), _p(dt), _ld(dt),
                                                 dropout[0], dropout[1], dropout[2], _stream()), 'lime_layernorm_bwd_dropout_f32')
        return dz, dgamma, dbeta, dzsum, dt
    check(lib.lime_layernorm_bwd_f32(_p(dy), _ld(dy), dy_div, dy_scale, _p(y), _ld(y), _p(gamma), _p(beta), _p(rstd), _p(dz), _ld(dz),
                                     M, E, _p(dgamma), _p(dbeta), _p(dzsum), 0, _p(ws), ws.numel(), _stream()),
          'lime_layernorm_bwd_f32')
    return dz, dgamma, dbeta, dzsum


def relu_bwd_(dh, h, scale=1.0):
    """dh = dh * scale where h > 0, else 0 -- in place."""
    lib = _lib.load()
    _mat(dh, 'dh')
    _mat(h, 'h')
    if dh.shape != h.shape:
        raise ValueError('dh and h differ in shape')
    check(lib.lime_relu_bwd_f32(_p(dh), _ld(dh), _p(h), _ld(h), dh.shape[0], dh.shape[1], scale, _stream()), 'lime_relu_bwd_f32')
    return dh


def token_attention_bwd(q, k, v, dout, n_seq, S, n_head, head_dim, scale, head_stride=None, dqkv=None, out=None,
                        dropout=None, key_mask=None, lse=None):
    """Backward of the unmasked ``token_attention``: q / k / v column views of one packed qkv buffer [tokens, 3 * n_head *
    head_stride]; returns dqkv in the same layout.  ``out``: the forward's result (needed for S > 128).  ``dropout``:
    (p, seed, site) of the ``token_attention_dropout`` forward."""
    lib = _lib.load()
    hs = head_dim if head_stride is None else head_stride
    W = n_head * hs
    for t, name in ((q, 'q'), (k, 'k'), (v, 'v')):
        _mat(t, name)
        if t.shape[0] != n_seq * S or t.shape[1] != W:
            raise ValueError('%s must be [n_seq * S, n_head * head_stride]' % name)
    if not (_ld(q) == _ld(k) == _ld(v)):
        raise ValueError('q, k, v must share one leading dimension')
    _mat(dout, 'dout')
    if tuple(dout.shape) != (n_seq * S, n_head * head_dim):
        raise ValueError('dout must be [n_seq * S, n_head * head_dim]')
    if dqkv is None:
        dqkv = torch.empty((n_seq * S, 3 * W), dtype=torch.float32, device=q.device)
    _mat(dqkv, 'dqkv')
    dq, dk, dv = dqkv[:, :W], dqkv[:, W:2 * W], dqkv[:, 2 * W:]
    ws, need = None, lib.lime_token_attention_bwd_workspace(n_seq, S, n_head)
    if need:
        if out is None:
            raise ValueError('S > 128 needs the forward output `out`')
        _mat(out, 'out')
        if tuple(out.shape) != tuple(dout.shape):
            raise ValueError('out must have the shape of dout')
        ws = _workspace(q.device, need)
    if lse is not None and need:                   # S > 128 with the forward's statistics: no Q K^T pass for them
        if dropout or key_mask is not None:
            raise ValueError('lse goes with the plain (no dropout, no key mask) forward')
        _vec(lse, 'lse', n_seq * S * n_head)
        check(lib.lime_token_attention_bwd_lse_f32(_p(q), _p(k), _p(v), _ld(q), _p(out), _ld(out), _p(lse), _p(dout), _ld(dout), _p(dq),
                                                   _p(dk), _p(dv), _ld(dqkv), n_seq, S, n_head, head_dim, hs, scale, _p(ws), ws.numel(),
                                                   _stream()), 'lime_token_attention_bwd_lse_f32')
        return dqkv
    check(lib.lime_token_attention_bwd_f32(_p(q), _p(k), _p(v), _ld(q), _p(out), _ld(out) if out is not None else 0, _p(dout),
                                           _ld(dout), _p(dq), _p(dk), _p(dv), _ld(dqkv), n_seq, S, n_head, head_dim, hs, scale,
                                           _p(ws), ws.numel() if ws is not None else 0, *(dropout or (0.0, 0, 0)),
                                           _p(_mask_u8(key_mask, 'key_mask')), _stream()), 'lime_token_attention_bwd_f32')
    return dqkv


def dropout2(src, p, seed, site1, site2, out=None):
    """dropout_site2(dropout_site1(src)) in one pass; ``out`` may be ``src``."""
    lib = _lib.load()
    _mat(src, 'src')
    if out is None:
        out = torch.empty(tuple(src.shape), dtype=torch.float32, device=src.device)
    _mat(out, 'out')
    if out.shape != src.shape:
        raise ValueError('out must have the shape of src')
    check(lib.lime_dropout2_f32(_p(src), _ld(src), _p(out), _ld(out), src.shape[0], src.shape[1], p, seed, site1, site2, _stream()),
          'lime_dropout2_f32')
    return out


def dropout(src, p, seed, site, out=None):
    """keep * src / (1 - p) with the counter-based mask of (seed, site); ``out`` may be ``src`` (in place)."""
    lib = _lib.load()
    _mat(src, 'src')
    if out is None:
        out = torch.empty(tuple(src.shape), dtype=torch.float32, device=src.device)
    _mat(out, 'out')
    if out.shape != src.shape:
        raise ValueError('out must have the shape of src')
    check(lib.lime_dropout_f32(_p(src), _ld(src), _p(out), _ld(out), src.shape[0], src.shape[1], p, seed, site, _stream()), 'lime_dropout_f32')
    return out


def embed_pe_dropout(ids, table, pe, period, p, seed, site_emb, site_pe):
    """drop(drop(table[ids]) + pe[r % period]) -> [len(ids), dim]."""
    lib = _lib.load()
    _vec(ids, 'ids', dtype=torch.int32)
    _mat(table, 'table')
    _mat(pe, 'pe')
    out = torch.empty((ids.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    check(lib.lime_embed_pe_dropout_f32(_p(ids), _p(table), _ld(table), _p(pe), _ld(pe), period, _p(out), _ld(out), ids.numel(),
                                        table.shape[1], p, seed, site_emb, site_pe, _stream()), 'lime_embed_pe_dropout_f32')
    return out


def dropout_add_layernorm(t, res, gamma, beta, eps, p, seed, site, want_rstd=True):
    """(LayerNorm(res + drop(t)), rstd)."""
    lib = _lib.load()
    _mat(t, 't')
    _mat(res, 'res')
    M, E = t.shape
    if res.shape != t.shape:
        raise ValueError('res must have the shape of t')
    _vec(gamma, 'gamma', E)
    _vec(beta, 'beta', E)
    y = torch.empty((M, E), dtype=torch.float32, device=t.device)
    rstd = torch.empty(M, dtype=torch.float32, device=t.device) if want_rstd else None
    check(lib.lime_dropout_add_layernorm_f32(_p(t), _ld(t), _p(res), _ld(res), _p(gamma), _p(beta), eps, _p(y), _ld(y), _p(rstd), M, E,
                                             p, seed, site, _stream()), 'lime_dropout_add_layernorm_f32')
    return y, rstd


def token_attention_dropout(q, k, v, n_seq, S, n_head, head_dim, scale, p, seed, site, head_stride=None):
    """Unmasked encoder attention with dropout on the probabilities (training mode); layouts as ``token_attention``."""
    lib = _lib.load()
    hs = head_dim if head_stride is None else head_stride
    for t, name in ((q, 'q'), (k, 'k'), (v, 'v')):
        _mat(t, name)
        if t.shape[0] != n_seq * S or t.shape[1] != n_head * hs:
            raise ValueError('%s must be [n_seq * S, n_head * head_stride]' % name)
    if not (_ld(q) == _ld(k) == _ld(v)):
        raise ValueError('q, k, v must share one leading dimension')
    out = torch.empty((n_seq * S, n_head * head_dim), dtype=torch.float32, device=q.device)
    need = lib.lime_token_attention_stats_workspace(n_seq, S, n_head)
    ws = _workspace(q.device, need) if need else None
    check(lib.lime_token_attention_dropout_f32(_p(q), _p(k), _p(v), _ld(q), _p(out), _ld(out), n_seq, S, n_head, head_dim, hs, scale,
                                               p, seed, site, _p(ws), ws.numel() if ws is not None else 0, _stream()),
          'lime_token_attention_dropout_f32')
    return out


DETERMINISTIC_EMBED_BWD = True      # word-table gradient by sort + segmented sum (no atomics); False: the float-atomic kernel


def embed_bwd(ids, dx, dtable, hot_id=0, accumulate=False):
    """dtable[ids[r]] = sum of dx[r] over the rows with that id, for a dtable the caller ZEROED (rows no id names are left alone).
    Tables of <= 32 rows: atomic-free LDS accumulation; larger tables with dim <= 320: stable sort of the positions by id +
    segmented sum in a fixed order (lime_embed_bwd_sorted_f32: bitwise reproducible) -- this back end STORES each touched row, the
    other two ADD into it, so a dtable that already holds a gradient needs ``accumulate=True`` (the sorted back end then sums into a
    zeroed scratch table and adds it: the same result on every back end); otherwise (or with DETERMINISTIC_EMBED_BWD off) float
    atomics."""
    lib = _lib.load()
    _vec(ids, 'ids', dtype=torch.int32)
    _mat(dx, 'dx')
    _mat(dtable, 'dtable')
    if dx.shape[0] != ids.numel() or dx.shape[1] != dtable.shape[1]:
        raise ValueError('dx must be [len(ids), dim]')
    if dtable.shape[0] <= 32:              # a handful of rows: per-column LDS accumulation instead of contended atomics
        check(lib.lime_embed_bwd_small_f32(_p(ids), _p(dx), _ld(dx), _p(dtable), _ld(dtable), ids.numel(), dx.shape[1],
                                           dtable.shape[0], _stream()), 'lime_embed_bwd_small_f32')
        return dtable
    if DETERMINISTIC_EMBED_BWD and dx.shape[1] <= 320 and ids.numel() > 0:
        if accumulate:
            return dtable.add_(embed_bwd(ids, dx, torch.zeros_like(dtable), hot_id))
        sorted_ids, order = torch.sort(ids, stable=True)
        order = order.to(torch.int32)
        need = int(lib.lime_embed_bwd_sorted_workspace(ids.numel(), dx.shape[1]))
        ws = torch.empty(need, dtype=torch.float32, device=dx.device)
        check(lib.lime_embed_bwd_sorted_f32(_p(order), _p(sorted_ids), _p(dx), _ld(dx), _p(dtable), _ld(dtable), ids.numel(), dx.shape[1],
                                            _p(ws), need, _stream()), 'lime_embed_bwd_sorted_f32')
        return dtable
    check(lib.lime_embed_bwd_f32(_p(ids), _p(dx), _ld(dx), _p(dtable), _ld(dtable), ids.numel(), dx.shape[1], hot_id, _stream()),
          'lime_embed_bwd_f32')
    return dtable


def grad_clip_coef(g, max_norm):
    """[norm, min(1, max_norm / (norm + 1e-6))] of the flat gradient buffer, as a 2-element device tensor."""
    lib = _lib.load()
    _vec(g, 'g')
    out = torch.empty(2, dtype=torch.float32, device=g.device)
    ws = _workspace(g.device, 1024)
    check(lib.lime_grad_clip_coef_f32(_p(g), g.numel(), float(max_norm), _p(out), _p(ws), ws.numel(), _stream()), 'lime_grad_clip_coef_f32')
    return out


def adam_step_(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=None):
    lib = _lib.load()
    n = p.numel()
    for t, name in ((p, 'p'), (g, 'g'), (m, 'm'), (v, 'v')):
        _vec(t, name, n)
    check(lib.lime_adam_f32(_p(p), _p(g), _p(m), _p(v), n, lr, betas[0], betas[1], eps, weight_decay, step, _p(grad_scale), _stream()),
          'lime_adam_f32')
    return p


def nll_softmax(logits, want_grad=True):
    """(loss [1], dlogits [B, K] or None) of trainer.py:71-73."""
    lib = _lib.load()
    _mat(logits, 'logits')
    B, K = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    d = torch.empty((B, K), dtype=torch.float32, device=logits.device) if want_grad else None
    check(lib.lime_nll_softmax_f32(_p(logits), _ld(logits), B, K, _p(loss), _p(d), _ld(d) if d is not None else K, _stream()),
          'lime_nll_softmax_f32')
    return loss, d


# ---- backward of the fused tail kernels -----------------------------------------------------------------------------------
def intent_fuse_bwd(intents, att_hidden, affine2_t, affine2_b, dcontent, M, k, D, A):
    """-> (d_intents [2Mk, D], d_hidden [2Mk, A], d_affine2_title [A], d_affine2_body [A])."""
    lib = _lib.load()
    _mat(intents, 'intents')
    _mat(att_hidden, 'att_hidden')
    _mat(dcontent, 'dcontent')
    if not (intents.is_contiguous() and att_hidden.is_contiguous()) or tuple(intents.shape) != (2 * M * k, D) or \
            tuple(att_hidden.shape) != (2 * M * k, A) or dcontent.shape[0] != M or dcontent.shape[1] < 2 * D:
        raise ValueError('intent_fuse_bwd: operand shapes')
    dev = intents.device
    d_int, d_hid = torch.empty_like(intents), torch.empty_like(att_hidden)
    da_t, da_b = torch.empty(A, dtype=torch.float32, device=dev), torch.empty(A, dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.lime_intent_fuse_bwd_workspace(M, A))
    check(lib.lime_intent_fuse_bwd_f32(_p(intents), _p(att_hidden), _p(_vec(affine2_t, 'affine2_t', A)), _p(_vec(affine2_b, 'affine2_b', A)),
                                       _p(dcontent), _ld(dcontent), _p(d_int), _p(d_hid), _p(da_t), _p(da_b), M, k, D, A, _p(ws),
                                       ws.numel(), _stream()), 'lime_intent_fuse_bwd_f32')
    return d_int, d_hid, da_t, da_b


def gate_ln_bwd(y, x, scale, bias, gamma, beta, eps, dout):
    """-> (dy, dx, dscale [rows], dbias, dgamma, dbeta) of ``gate_ln``; y / x / dout: contiguous [rows, D]."""
    lib = _lib.load()
    D = x.shape[-1]
    x = _vec(x.contiguous(), 'x')
    rows = x.numel() // D
    y = _vec(y.contiguous(), 'y', x.numel())
    dout = _vec(dout.contiguous(), 'dout', x.numel())
    dev = x.device
    dy, dx = torch.empty_like(x), torch.empty_like(x)
    dscale = torch.empty(rows, dtype=torch.float32, device=dev)
    dbias, dgamma, dbeta = (torch.empty(D, dtype=torch.float32, device=dev) for _ in range(3))
    ws = _workspace(dev, lib.lime_gate_ln_bwd_workspace(rows, D))
    check(lib.lime_gate_ln_bwd_f32(_p(y), _p(x), _p(_vec(scale.contiguous(), 'scale', rows)), _p(_vec(bias, 'bias', D)),
                                   _p(_vec(gamma, 'gamma', D)), _p(_vec(beta, 'beta', D)), eps, _p(dout), _p(dy), _p(dx), _p(dscale),
                                   _p(dbias), _p(dgamma), _p(dbeta), rows, D, _p(ws), ws.numel(), _stream()), 'lime_gate_ln_bwd_f32')
    return dy, dx, dscale, dbias, dgamma, dbeta


def interest_match_bwd(kp, qp, g, cand, remaining, dlogits, B, N, H, A, D, scale, alpha, beta, use_weight, use_penalty):
    """-> (dkp [B*H, A], dqp [B*N, A], dg [B*H, D], dcand [B*N, D])."""
    lib = _lib.load()
    _vec(kp, 'kp', B * H * A)
    _vec(qp, 'qp', B * N * A)
    _vec(g, 'g', B * H * D)
    _vec(cand, 'cand', B * N * D)
    dlogits = _vec(dlogits.contiguous(), 'dlogits', B * N)
    if use_weight:
        remaining = _vec(remaining.contiguous(), 'remaining', B * N)
    dev = kp.device
    dkp = torch.empty((B * H, A), dtype=torch.float32, device=dev)
    dqp = torch.empty((B * N, A), dtype=torch.float32, device=dev)
    dg = torch.empty((B * H, D), dtype=torch.float32, device=dev)
    dcand = torch.empty((B * N, D), dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.lime_interest_match_bwd_workspace(B, N, H, A, D))
    check(lib.lime_interest_match_bwd_f32(_p(kp), _p(qp), _p(g), _p(cand), _p(remaining) if use_weight else None, _p(dlogits), _p(dkp),
                                          _p(dqp), _p(dg), _p(dcand), B, N, H, A, D, scale, alpha, beta, int(use_weight),
                                          int(use_penalty), _p(ws), ws.numel(), _stream()), 'lime_interest_match_bwd_f32')
    return dkp, dqp, dg, dcand


def cand_attn_weights_train(qp, kp, mask, B, N, H, D, n_head, p=0.0, seed=0, site=0):
    """agg [B, H] of layers.py:66-81 with the training-mode dropout on the per-head probabilities."""
    lib = _lib.load()
    _vec(qp, 'qp', B * N * D)
    _vec(kp, 'kp', B * H * D)
    m = _mask_u8(mask, 'mask')
    if m.numel() != B * H:
        raise ValueError('mask must be [B, H]')
    agg = torch.empty((B, H), dtype=torch.float32, device=qp.device)
    check(lib.lime_cand_attn_weights_train_f32(_p(qp), _p(kp), _p(m), _p(agg), B, N, H, D, n_head, p, seed, site, _stream()),
          'lime_cand_attn_weights_train_f32')
    return agg


def cand_attn_weights_bwd(qp, kp, mask, dagg, B, N, H, D, n_head, p=0.0, seed=0, site=0):
    """-> (dqp [B*N, D], dkp [B*H, D])."""
    lib = _lib.load()
    _vec(qp, 'qp', B * N * D)
    _vec(kp, 'kp', B * H * D)
    m = _mask_u8(mask, 'mask')
    dagg = _vec(dagg.contiguous(), 'dagg', B * H)
    dqp = torch.empty((B * N, D), dtype=torch.float32, device=qp.device)
    dkp = torch.empty((B * H, D), dtype=torch.float32, device=qp.device)
    check(lib.lime_cand_attn_weights_bwd_f32(_p(qp), _p(kp), _p(m), _p(dagg), _p(dqp), _p(dkp), B, N, H, D, n_head, p, seed, site, _stream()),
          'lime_cand_attn_weights_bwd_f32')
    return dqp, dkp
