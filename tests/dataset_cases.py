"""The toy corpus and behaviour indices behind tests/golden/dataset_*.npz (shared by the generator and the tests)."""
from lime_cikm25_amd import make_config, synth

SAMPLING_SEED = 1234
TRAIN_INDICES = [0, 3, 4, 9, 17, 22, 31, 39]
DEV_INDICES = [0, 1, 5, 8, 13, 29]
TEST_INDICES = [0, 2, 7, 14]


def build():
    cfg = make_config(max_history_num=6, max_title_length=8, max_abstract_length=12, vocabulary_size=300, negative_sample_num=4)
    return cfg, synth.synth_corpus(cfg, n_news=60, n_train=40, n_dev=30, seed=11)
