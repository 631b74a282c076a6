"""From the raw tsv files to metrics on the MI355X, every stage on this repo's path: formats.build_corpus (parsers pinned by
tests/golden/formats.json) -> device-side batch assembly -> TrainStep (forward, backward, clip, Adam) -> eval-mode scoring of
the dev rows -> rank file -> AUC / MRR / nDCG (evaluate.scoring).  A plumbing test on a 12-news toy dataset: the loss of the
training batches must fall and the metrics must be well formed."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR
from lime_cikm25_amd import DeviceBehaviors, DeviceCorpus, Model, formats, make_config, util
from lime_cikm25_amd.device_data import negative_sampling
from lime_cikm25_amd.training import TrainStep

pytestmark = pytest.mark.gpu


def test_files_to_metrics(tmp_path):
    g = json.load(open(os.path.join(GOLDEN_DIR, 'formats.json')))
    L = g['lines']
    cfg = make_config(max_history_num=g['max_history_num'], max_title_length=g['max_title_length'],
                      max_abstract_length=g['max_abstract_length'], vocabulary_size=len(g['word_dict']), negative_sample_num=2,
                      category_num=len(g['category_dict']) + 1, subCategory_num=len(g['subCategory_dict']) + 1,
                      user_num=len(g['user_ID_dict']), batch_size=16)
    corpus = formats.build_corpus(cfg, [L['train_news'], L['dev_news'], L['test_news']],
                                  [L['train_behaviors'], L['dev_behaviors'], L['test_behaviors']], g['news_ID_dict'],
                                  g['user_ID_dict'], g['category_dict'], g['subCategory_dict'], g['word_dict'], dataset='adressa')
    dc = DeviceCorpus(corpus)
    np.random.seed(3)
    train = DeviceBehaviors.from_train(dc, corpus, *negative_sampling(corpus.train_behaviors, cfg.negative_sample_num))
    dev = DeviceBehaviors.from_devtest(dc, corpus, 'dev')

    torch.manual_seed(0)
    model = Model(cfg)
    model.initialize()
    torch.nn.init.normal_(model.news_encoder.base_news_encoder.word_embedding.weight, std=0.1)
    model = model.cuda().train()
    step = TrainStep(model, lr=1e-3, gradient_clip_norm=4.0)
    rows = list(range(train.num))
    losses = []
    for _ in range(30):
        batch = train.assemble(rows)
        losses.append(float(step.step(*batch, batch[24] - batch[23])))            # remaining lifetime: trainer.py:126-127
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses

    truth = tmp_path / 'truth.txt'
    with open(truth, 'w') as f:
        for i, labels in enumerate(formats.truth_labels(L['dev_behaviors'])):
            f.write('%d %s\n' % (i + 1, json.dumps(labels).replace(' ', '')))
    batches = [dev.assemble(list(range(dev.num)))]
    auc, mrr, ndcg5, ndcg10 = util.compute_scores(model, batches, corpus.dev_indices, str(tmp_path / 'rank.txt'), str(truth))
    for m in (auc, mrr, ndcg5, ndcg10):
        assert 0.0 <= m <= 1.0
    ranks = [json.loads(line.split(' ', 1)[1]) for line in open(tmp_path / 'rank.txt')]
    assert [sorted(r) for r in ranks] == [list(range(1, len(lab) + 1)) for lab in formats.truth_labels(L['dev_behaviors'])]
