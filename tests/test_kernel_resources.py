"""Register / scratch budget of every kernel in liblime_hip.so, from the remarks hipcc prints while the library is built
(`-Rpass-analysis=kernel-resource-usage`, summarised by lime_cikm25_amd.build into liblime_hip.resources.json).  CPU only.

Why this is a test and not a note: the kernels that synchronise an LDS-DMA ring with COUNTED `s_waitcnt vmcnt(N)` (csrc/lds_dma.h
wait_vm<N>, csrc/gemm_mid_f32.hip) were written -- and their counts derived -- for an instruction stream without compiler-made
vector-memory operations.  A register spill puts scratch_store / scratch_load into that stream: they count in the same vmcnt, every
reload is followed by a compiler `s_waitcnt vmcnt(0)` that drains the ring, and one build of the bf16 block with 11
spilled registers returned wrong lanes (profiles/r02_notes.md).  So: no scratch at all in those kernels, and nowhere else more than
what is recorded below -- a compiler or source change that brings spills back fails here, on the CPU, before anything runs.
"""
import glob
import json
import os
import re

import pytest

from lime_cikm25_amd import build

CSRC = build.CSRC

# translation units whose kernels use counted vmcnt waits (N > 0): scratch-free, every kernel
COUNTED_VMCNT_UNITS = ('ffn_bf16', 'inproj_bf16', 'gemm_mid_f32', 'gemm_group_f32')

# kernels that are known to spill today: regex on the demangled name -> scratch bytes per lane allowed (their waits are all
# vmcnt(0), where a scratch access costs a drain, never an early read).  Everything else must be scratch-free.
KNOWN_SCRATCH = {
    r'^token_attn_bwd_kernel<128>': 16,
    r'^wgrad_kernel<5, false>': 16,
    r'^gemm_f32_kernel<1, 4, 4, 2, 1, true, false, 0, false, false>': 320,      # unaligned LayerNorm fallback (rarely taken)
    r'^gemm_f32_kernel<1, 5, 4, 2, 1, true, false, 0, false, false>': 960,
    # residual-into-accumulator LayerNorm GEMMs (tile-level code: accumulator init / epilogue; no scratch access inside the chunk loop)
    r'^gemm_pp_kernel<10, true, false, 2, false, false, (true|false), false, [01]>': 96,
    r'^gemm_pp_kernel<10, true, false, [23], true, (true|false), false, false, [01]>': 136,
    # split-product GEMM (csrc/gemm_sp_f32.hip, waits are all vmcnt(0)): 160 accumulators + 60 fragment registers of 256 -- the tile
    # boundary (residual loads into the accumulators, LayerNorm epilogue) spills, the chunk loop does not (ISA checked, profiles/r03_notes.md)
    r'^gemm_sp_kernel<10, (true|false), (true|false), [01], ': 192,
    r'^gemm_sp_kernel<10, false, false, 2, ': 192,
    r'^gemm_sp_kernel<8, false, false, 1, false, false, true>': 16,
    r'^gemm_sp_kernel<10, true, false, 2, ': 1040,       # off by default (lime_set_split_gemm(3))
}


@pytest.fixture(scope='module')
def resources():
    build.build_library()
    want = build.source_hash()
    try:
        res = json.load(open(build.RESOURCES))
    except OSError:
        res = {}
    if res.get('source_hash') != want:               # a library built before the summary existed, or by hand
        build.build_library(force=True)
        res = json.load(open(build.RESOURCES))
    assert res['source_hash'] == want
    return res['units']


def _uses_counted_vmcnt(text):
    """True when the unit waits on a vmcnt other than 0 (wait_vm<N> with N != 0, or a templated / literal vmcnt(N > 0))."""
    text = re.sub(r'//[^\n]*', '', text)
    for m in re.finditer(r'wait_vm<([^>]*)>', text):
        if m.group(1).strip() != '0':
            return True
    for m in re.finditer(r's_waitcnt vmcnt\(([^)]*)\)', text):
        if m.group(1).strip() != '0':
            return True
    return False


def test_every_unit_with_counted_vmcnt_is_listed():
    units = set()
    for path in glob.glob(os.path.join(CSRC, '*.hip')):
        if _uses_counted_vmcnt(open(path).read()):
            units.add(os.path.splitext(os.path.basename(path))[0])
    assert units <= set(COUNTED_VMCNT_UNITS), 'counted vmcnt waits in a unit this test does not guard: %s' % sorted(units - set(COUNTED_VMCNT_UNITS))


def test_counted_vmcnt_kernels_are_scratch_free(resources):
    bad = []
    for unit in COUNTED_VMCNT_UNITS:
        for name, r in resources.get(unit, {}).items():
            if r['scratch'] or r['vgpr_spill']:
                bad.append('%s: %s scratch %d B/lane, %d VGPRs spilled' % (unit, name[:100], r['scratch'], r['vgpr_spill']))
    assert not bad, 'kernels with counted vmcnt waits must not spill to scratch:\n' + '\n'.join(bad)


def test_sgpr_lane_spills_stay_bounded(resources):
    """SGPR spills go to VGPR lanes (v_writelane / v_readlane, no memory traffic, nothing in vmcnt): legal, but each costs VALU issue
    slots in MFMA-paced loops, and the one wrong-lanes build (profiles/r02_notes.md) corrupted exactly four lanes (12-15 = four
    spill slots) of a register.  Today's counts are recorded; growth fails."""
    limits = {r'^ffn_bf16_kernel<(true|false), true>': 215, r'^ffn_bf16_kernel<(true|false), false>': 95, r'^inproj_bf16_kernel': 100}
    bad = []
    for unit in COUNTED_VMCNT_UNITS:
        for name, r in resources.get(unit, {}).items():
            lim = max([v for k, v in limits.items() if re.search(k, name)], default=0)
            if r['sgpr_spill'] > lim:
                bad.append('%s: %s %d SGPRs spilled to lanes (limit %d)' % (unit, name[:100], r['sgpr_spill'], lim))
    assert not bad, '\n'.join(bad)


def test_no_new_spills(resources):
    bad, seen = [], 0
    for unit, kernels in resources.items():
        for name, r in kernels.items():
            seen += 1
            if not r['scratch']:
                continue
            allowed = max([v for k, v in KNOWN_SCRATCH.items() if re.search(k, name)], default=0)
            if r['scratch'] > allowed:
                bad.append('%s: %s scratch %d B/lane (allowed %d)' % (unit, name[:110], r['scratch'], allowed))
    assert seen > 100, 'the resource summary looks empty (%d kernels)' % seen
    assert not bad, 'new or grown register spills:\n' + '\n'.join(bad)


def test_summary_covers_every_unit(resources):
    units = {os.path.splitext(os.path.basename(p))[0] for p in build.sources() if p.endswith('.hip')}
    assert units <= set(resources), 'units without a resource summary: %s' % sorted(units - set(resources))
