"""Generate tests/golden/dataset_{train,dev,test}.npz by running the IMPORTED REFERENCE datasets (dataset.py) on the
synthetic toy corpus of lime_cikm25_amd.synth.synth_corpus (build container only).

    python tools/make_dataset_goldens.py

Train_Dataset.negative_sampling draws from numpy's global generator: it is seeded here and the sampled candidate tables
(train_samples / train_freshness / train_user_topic_lifetime) are stored too -- they are INPUTS of the device-side
assembly (negative sampling itself stays on the host, as in the reference).  Outputs: the default-collated 25-tuples of
``__getitem__`` for a fixed list of behaviour indices.
"""
import os
import sys

import numpy as np
import torch
from torch.utils.data import default_collate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import ref_harness  # noqa: E402
from lime_cikm25_amd import synth  # noqa: E402
import dataset_cases  # noqa: E402


def main():
    ref_harness.import_reference()
    import dataset as ref_dataset                                           # /root/reference/dataset.py
    cfg, corpus = dataset_cases.build()
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    train = ref_dataset.Train_Dataset(corpus)
    np.random.seed(dataset_cases.SAMPLING_SEED)
    train.negative_sampling()
    batch = default_collate([train[i] for i in dataset_cases.TRAIN_INDICES])
    store = {'out%02d' % k: (t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)) for k, t in enumerate(batch)}
    store['train_samples'] = np.asarray(train.train_samples, dtype=np.int64)
    store['train_freshness'] = np.asarray(train.train_freshness, dtype=np.float64)
    store['train_user_topic_lifetime'] = np.asarray(train.train_user_topic_lifetime, dtype=np.float64)
    np.savez_compressed(os.path.join(out_dir, 'dataset_train.npz'), **store)
    print('train: %d outputs, shapes %s' % (len(batch), [tuple(np.asarray(v).shape) for v in store.values()][:6]))
    for mode, idx in (('dev', dataset_cases.DEV_INDICES), ('test', dataset_cases.TEST_INDICES)):
        ds = ref_dataset.DevTest_Dataset(corpus, mode)
        batch = default_collate([ds[i] for i in idx])
        store = {'out%02d' % k: (t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)) for k, t in enumerate(batch)}
        np.savez_compressed(os.path.join(out_dir, 'dataset_%s.npz' % mode), **store)
        print('%s: %d outputs' % (mode, len(batch)))


if __name__ == '__main__':
    main()
