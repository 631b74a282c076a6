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
    # the cached dev pass (every news through the token encoders once) over the same single batch: the same rank file
    cached = util.compute_scores_cached(model, dev, corpus.dev_indices, str(tmp_path / 'rank_cached.txt'), str(truth), rows_per_forward=dev.num)
    assert open(tmp_path / 'rank_cached.txt').read() == open(tmp_path / 'rank.txt').read()
    assert cached == (auc, mrr, ndcg5, ndcg10)
    ranks = [json.loads(line.split(' ', 1)[1]) for line in open(tmp_path / 'rank.txt')]
    assert [sorted(r) for r in ranks] == [list(range(1, len(lab) + 1)) for lab in formats.truth_labels(L['dev_behaviors'])]
    # the other two ways the reference derives the remaining lifetime (util.py:98-106): the cached pass follows config.lifetime_type
    # exactly as compute_scores does
    for lt in ('fixed', 'topic_wise'):
        cfg.lifetime_type = lt
        cfg.fixed_lifetime = 5 * 3600
        cfg.category_lifetime_map = torch.linspace(600.0, 9e4, cfg.category_num)
        a = util.compute_scores(model, batches, corpus.dev_indices, str(tmp_path / ('rank_%s.txt' % lt)), str(truth))
        b = util.compute_scores_cached(model, dev, corpus.dev_indices, str(tmp_path / ('rank_cached_%s.txt' % lt)), str(truth), rows_per_forward=dev.num)
        assert open(tmp_path / ('rank_cached_%s.txt' % lt)).read() == open(tmp_path / ('rank_%s.txt' % lt)).read(), lt
        assert a == b
    cfg.lifetime_type = 'user_topic'


def test_trainer_loop_selects_and_saves_the_best_epoch(tmp_path):
    """trainer.Trainer (the reference's Trainer.train, trainer.py:84-244) for three epochs on the toy dataset: dev metrics per
    epoch, the improving epochs checkpointed in the reference's file layout, the best one copied to best_model_dir."""
    from lime_cikm25_amd.trainer import Trainer
    from lime_cikm25_amd.training import load_checkpoint
    g = json.load(open(os.path.join(GOLDEN_DIR, 'formats.json')))
    L = g['lines']
    d = str(tmp_path)
    cfg = make_config(max_history_num=g['max_history_num'], max_title_length=g['max_title_length'],
                      max_abstract_length=g['max_abstract_length'], vocabulary_size=len(g['word_dict']), negative_sample_num=2,
                      category_num=len(g['category_dict']) + 1, subCategory_num=len(g['subCategory_dict']) + 1,
                      user_num=len(g['user_ID_dict']), batch_size=4, epoch=3, lr=1e-3, dataset='adressa',
                      model_dir=d + '/models', best_model_dir=d + '/best', dev_res_dir=d + '/dev', result_dir=d + '/results')
    corpus = formats.build_corpus(cfg, [L['train_news'], L['dev_news'], L['test_news']],
                                  [L['train_behaviors'], L['dev_behaviors'], L['test_behaviors']], g['news_ID_dict'],
                                  g['user_ID_dict'], g['category_dict'], g['subCategory_dict'], g['word_dict'], dataset='adressa')
    truth = tmp_path / 'truth.txt'
    with open(truth, 'w') as f:
        for i, labels in enumerate(formats.truth_labels(L['dev_behaviors'])):
            f.write('%d %s\n' % (i + 1, json.dumps(labels).replace(' ', '')))
    torch.manual_seed(0)
    np.random.seed(0)
    model = Model(cfg)
    model.initialize()
    torch.nn.init.normal_(model.news_encoder.base_news_encoder.word_embedding.weight, std=0.1)
    trainer = Trainer(model.cuda(), cfg, corpus, run_index=1, truth_file=str(truth))
    best = trainer.train()
    assert 1 <= best <= 3 and len(trainer.results['auc']) == 3 and all(0.0 <= v <= 1.0 for v in trainer.results['auc'])
    best_file = os.path.join(d, 'best', '#1', model.model_name)
    assert os.path.exists(best_file) and os.path.exists(os.path.join(d, 'results', '#1-dev'))
    log = open(os.path.join(d, 'dev', '#1', '%s-adressa-dev_log.txt' % model.model_name)).read().splitlines()
    assert log[0] == 'Epoch\tAUC\tMRR\tnDCG@5\tnDCG@10' and len(log) == 4
    fresh = Model(cfg).cuda()
    payload = load_checkpoint(best_file, fresh)                      # the reference's main.py:45 reads the same key
    assert model.model_name in payload


def test_trainer_dev_pass_fits_the_node_slots_when_batch_exceeds_history(tmp_path):
    """config.batch_size (8) > max_history_num (4), as at the reference's own settings (64 > 50): the dev pass scores
    config.batch_size rows per forward (trainer.py:153) -- twice that would exceed the H + batch_size GraphSAGE node slots
     -- and the cached and the re-encoding dev pass write the same rank file.  No truth file is passed: the Trainer
    writes it from the corpus's dev labels, in the reference's format (config.py:262-276)."""
    from lime_cikm25_amd.trainer import Trainer
    g = json.load(open(os.path.join(GOLDEN_DIR, 'formats.json')))
    L = g['lines']
    d = str(tmp_path)
    cfg = make_config(max_history_num=g['max_history_num'], max_title_length=g['max_title_length'],
                      max_abstract_length=g['max_abstract_length'], vocabulary_size=len(g['word_dict']), negative_sample_num=2,
                      category_num=len(g['category_dict']) + 1, subCategory_num=len(g['subCategory_dict']) + 1,
                      user_num=len(g['user_ID_dict']), batch_size=8, epoch=1, lr=1e-3, dataset='adressa',
                      model_dir=d + '/models', best_model_dir=d + '/best', dev_res_dir=d + '/dev/res', result_dir=d + '/results')
    assert cfg.batch_size > cfg.max_history_num
    corpus = formats.build_corpus(cfg, [L['train_news'], L['dev_news'], L['test_news']],
                                  [L['train_behaviors'], L['dev_behaviors'], L['test_behaviors']], g['news_ID_dict'],
                                  g['user_ID_dict'], g['category_dict'], g['subCategory_dict'], g['word_dict'], dataset='adressa')
    assert len(corpus.dev_indices) > 2 * cfg.batch_size - 1 > cfg.max_history_num + cfg.batch_size        # 2x would not fit
    torch.manual_seed(0)
    np.random.seed(0)
    model = Model(cfg)
    model.initialize()
    trainer = Trainer(model.cuda(), cfg, corpus, run_index=2)
    truth = os.path.join(d, 'dev', 'ref', 'truth-adressa.txt')
    assert trainer.truth_file == truth and os.path.exists(truth)
    assert trainer.train() == 1
    cached = trainer.evaluate(7)
    trainer.cached_eval = False
    plain = trainer.evaluate(8)
    name = model.model_name
    assert open(os.path.join(trainer.dev_res_dir, '%s-7.txt' % name)).read() == open(os.path.join(trainer.dev_res_dir, '%s-8.txt' % name)).read()
    assert cached == plain and all(0.0 <= v <= 1.0 for v in cached)
    with pytest.raises(ValueError):
        util.compute_scores_cached(model, trainer.dev, corpus.dev_indices, os.path.join(d, 'x.txt'), truth, rows_per_forward=2 * cfg.batch_size)
    with pytest.raises(ValueError):                                     # no labels, no file: refused before any training
        del corpus.dev_labels
        Trainer(model, cfg, corpus, run_index=3)
