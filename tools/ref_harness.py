"""Import the reference implementation (read-only at /root/reference) on CPU.

Only used in the build container to validate oracle/ and to generate tests/golden/*.npz;
nothing here runs on the GPU box (the reference does not travel).  Four third-party modules the
reference imports are absent from this image and cannot be installed:
``nltk`` / ``torchtext`` (preprocessing only), ``torch_scatter`` (SUE only) and
``torch_geometric``.  The first three get empty stand-ins.  ``torch_geometric.nn.GraphSAGE`` IS on
the scoring path (userEncoders.py:54-58,153); its stand-in below restates PyG's documented
``SAGEConv`` (mean aggregation over ``edge_index`` applied along node_dim=-2, ``lin_l`` with bias on
the aggregate, ``lin_r`` without bias on the root) -- PyG's version is unpinned upstream, so parity
at this one boundary is *unpinned* and DESIGN.md says so.
"""
import os
import pickle
import sys
import tempfile
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = '/root/reference'


class _SAGEConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index):
        src, dst = edge_index[0], edge_index[1]
        msg = x.index_select(-2, src)                                   # x_j for every edge j -> i
        agg = torch.zeros_like(x)
        agg.index_add_(-2, dst, msg)
        deg = torch.zeros(x.shape[-2], dtype=x.dtype, device=x.device)
        deg.index_add_(0, dst, torch.ones_like(dst, dtype=x.dtype))
        agg = agg / deg.clamp(min=1).unsqueeze(-1)
        return self.lin_l(agg) + self.lin_r(x)


class _GraphSAGE(nn.Module):
    """GraphSAGE(num_layers=1, out_channels given, jk=None): one SAGEConv, no act/dropout after it."""

    def __init__(self, in_channels, hidden_channels, num_layers, out_channels=None, dropout=0.0, **kw):
        super().__init__()
        assert num_layers == 1
        self.convs = nn.ModuleList([_SAGEConv(in_channels, out_channels or hidden_channels)])

    def forward(self, x, edge_index):
        return self.convs[0](x, edge_index)


class _Dummy(nn.Module):
    def __init__(self, *a, **kw):
        super().__init__()


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if 'torch_geometric' not in sys.modules:
        tg = mod('torch_geometric')
        tg.nn = mod('torch_geometric.nn', SAGEConv=_SAGEConv, GraphSAGE=_GraphSAGE, GCN=_Dummy,
                    LightGCN=_Dummy, LGConv=_Dummy)
    if 'torch_scatter' not in sys.modules:
        mod('torch_scatter', scatter_sum=None, scatter_softmax=None)
    if 'nltk' not in sys.modules:
        nl = mod('nltk')
        nl.tokenize = mod('nltk.tokenize', word_tokenize=None)
    if 'torchtext' not in sys.modules:
        tt = mod('torchtext')
        tt.vocab = mod('torchtext.vocab', GloVe=None)


def import_reference():
    """Returns the reference's ``model`` module (and leaves its siblings importable)."""
    sys.dont_write_bytecode = True
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import model as ref_model  # noqa: E402
    return ref_model


def build_reference_model(config, word_embedding):
    """``Model(config)`` of the reference; the word table is handed over through the pickle the
    reference opens from the cwd (newsEncoders.py:173-174)."""
    ref_model = import_reference()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        fn = 'word_embedding-%s-%s-%s-%s-%s-%s.pkl' % (
            config.word_threshold, config.word_embedding_dim, config.tokenizer, config.max_title_length,
            config.max_abstract_length, config.dataset)
        with open(os.path.join(tmp, fn), 'wb') as f:
            pickle.dump(word_embedding, f)
        os.chdir(tmp)
        try:
            m = ref_model.Model(config)
        finally:
            os.chdir(cwd)
    return m
