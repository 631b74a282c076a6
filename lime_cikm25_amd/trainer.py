"""The reference's training loop (trainer.py:17-244: ``Trainer.train``) on this repo's pieces: per epoch a fresh negative
sampling, shuffled batches assembled on the device (DeviceBehaviors), the native step (TrainStep: forward, backward, one
gradient all-reduce under torch.distributed, clip, Adam), then the dev pass (util.compute_scores) with model selection on
``config.dev_criterion``, a checkpoint of every improving epoch ({model_name: state_dict}, trainer.py:220) and early stopping.

Differences kept small and stated: batches come from the HBM-resident corpus instead of a DataLoader; under torch.distributed
rank r trains on rows r, r + W, ... of the epoch's permutation (DistributedSampler's rule, trainer.py:293-295) and only rank 0
evaluates and writes files; the optimizer state is saved next to the weights so that a run can resume.
"""
import os
import shutil

import numpy as np
import torch

from . import distributed, formats, util
from .device_data import DeviceBehaviors, DeviceCorpus, negative_sampling
from .training import TrainStep, save_checkpoint

_CRITERIA = ('auc', 'mrr', 'ndcg5', 'ndcg10')


def _mkdir(path):
    os.makedirs(path, exist_ok=True)
    return path


class Trainer:
    def __init__(self, model, config, corpus, run_index=0, truth_file=None, device_corpus=None, cached_eval=True):
        """``truth_file``: the dev truth file of config.py:262-276 ("<impression> [labels]" lines).  Without one it is written
        (as the reference's Config does at start-up) to ``<dev_res_dir>/../ref/truth-<dataset>.txt`` from ``corpus.dev_labels``
        (formats.build_corpus attaches them); a corpus without labels and no file is refused here, before any training.
        ``cached_eval``: the dev pass encodes every news once (util.compute_scores_cached) instead of once per row and slot; same
        scores.

        Under torch.distributed (WORLD_SIZE in the environment) the process group is initialised and the device selected
        FIRST, so that TrainStep's broadcast of rank 0's parameters really runs (DistributedDataParallel does that at
        construction, trainer.py:256), and each rank steps on ``config.batch_size // world`` rows (trainer.py:254-255)."""
        self.model, self.config, self.corpus, self.run_index = model, config, corpus, run_index
        self.rank, self.world = 0, 1
        if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) > 1:
            dev = next(model.parameters()).device
            if dev.type == 'cuda':
                local = int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count())
                torch.cuda.set_device(local)
                if dev.index not in (None, local):
                    raise ValueError('rank with LOCAL_RANK %d was handed a model on %s: move it to cuda:%d' % (local, dev, local))
            self.rank, self.world = distributed.init()
        if config.batch_size % self.world:
            raise ValueError('config.batch_size (%d) must be divisible by the world size (%d) (config.py:205)' % (config.batch_size, self.world))
        self.epoch = config.epoch
        self.batch_size = config.batch_size // self.world                      # rows per rank and step (trainer.py:254)
        self.eval_batch_size = config.batch_size                               # the dev pass of trainer.py:153 (rank 0 alone)
        self.dev_criterion, self.early_stopping_epoch = config.dev_criterion, config.early_stopping_epoch
        self.model_dir = _mkdir(os.path.join(config.model_dir, '#%d' % run_index))
        self.best_model_dir = _mkdir(os.path.join(config.best_model_dir, '#%d' % run_index))
        self.dev_res_dir = _mkdir(os.path.join(config.dev_res_dir, '#%d' % run_index))
        self.result_dir = _mkdir(config.result_dir)
        if truth_file is None:
            labels = getattr(corpus, 'dev_labels', None)
            if labels is None:
                raise ValueError('Trainer needs the dev truth file (config.py:262-276): pass truth_file=... or a corpus that carries '
                                 'dev_labels (formats.build_corpus attaches them)')
            truth_file = os.path.join(os.path.dirname(os.path.normpath(config.dev_res_dir)) or '.', 'ref', 'truth-%s.txt' % config.dataset)
            if self.rank == 0:
                formats.write_truth_file(truth_file, labels)
        self.truth_file = truth_file
        if getattr(config, 'dropout_rate', 0.0) == 0.0 and self.rank == 0:
            print('Trainer: config.dropout_rate is 0 -- the reference trains with 0.2 (config.py:78); pass dropout_rate=0.2 for its recipe')
        self.cached_eval = cached_eval and config.lifetime_type == 'user_topic' and config.fusion_method == 'concat'
        self.dc = device_corpus if device_corpus is not None else DeviceCorpus(corpus)
        self.dev = DeviceBehaviors.from_devtest(self.dc, corpus, 'dev')
        self.step = TrainStep(model, lr=config.lr, weight_decay=config.weight_decay, gradient_clip_norm=config.gradient_clip_norm)
        self.results = {k: [] for k in _CRITERIA}
        self.best_dev_epoch, self.best, self.epoch_not_increase = 0, -1.0, 0

    def _remaining(self, batch):
        cfg = self.config                                                    # trainer.py:121-129
        if cfg.lifetime_type == 'fixed':
            return cfg.fixed_lifetime - batch[23]
        if cfg.lifetime_type == 'topic_wise':
            return cfg.category_lifetime_map.to(batch[15].device)[batch[15].long()] - batch[23]
        if cfg.lifetime_type == 'user_topic':
            return batch[24] - batch[23]
        raise ValueError('Invalid lifetime_type')

    def train_epoch(self, e):
        cfg, model = self.config, self.model
        samples = negative_sampling(self.corpus.train_behaviors, cfg.negative_sample_num)     # dataset.py:42-77
        train = DeviceBehaviors.from_train(self.dc, self.corpus, *samples)
        # shuffle (DataLoader(shuffle=True), trainer.py:86) with a generator every rank seeds alike, as DistributedSampler
        # does with (seed, epoch): the ranks' row sets must partition ONE permutation
        order = np.random.RandomState(getattr(cfg, 'seed', 0) + e).permutation(train.num)
        rows = distributed.sampler_rows(train.num, self.rank, self.world, order)
        model.train()
        total, seen = 0.0, 0
        for i in range(0, len(rows), self.batch_size):
            chunk = [int(r) for r in rows[i:i + self.batch_size]]
            batch = train.assemble(chunk)
            loss = self.step.step(*batch, self._remaining(batch))
            total += float(loss) * len(chunk)
            seen += len(chunk)
        return total / max(1, seen)

    def evaluate(self, e):
        """The dev pass of trainer.py:153: ``compute_scores(model, corpus, self.batch_size, 'dev', ...)`` -- config.batch_size rows per
        forward.  The row count of a forward is the GraphSAGE source count, so it decides the scores, and it must
        not exceed the H + config.batch_size node slots (the 2 x batch size of main.py:50 does, at the reference's own
        batch_size = 64 / max_history_num = 50)."""
        out = os.path.join(self.dev_res_dir, '%s-%d.txt' % (self.model.model_name, e))
        per = self.eval_batch_size
        if self.cached_eval:
            return util.compute_scores_cached(self.model, self.dev, self.corpus.dev_indices, out, self.truth_file, per)
        rows = list(range(self.dev.num))
        batches = (self.dev.assemble(rows[i:i + per]) for i in range(0, len(rows), per))
        return util.compute_scores(self.model, batches, self.corpus.dev_indices, out, self.truth_file)

    def train(self):
        name = self.model.model_name
        for e in range(1, self.epoch + 1):
            loss = self.train_epoch(e)
            print('Epoch %d : train done\nloss = %.6f' % (e, loss))
            if self.rank == 0:
                auc, mrr, ndcg5, ndcg10 = self.evaluate(e)
                for k, v in zip(_CRITERIA, (auc, mrr, ndcg5, ndcg10)):
                    self.results[k].append(v)
                print('Epoch %d : dev done\nAUC = %.4f\nMRR = %.4f\nnDCG@5 = %.4f\nnDCG@10 = %.4f' % (e, auc, mrr, ndcg5, ndcg10))
                # 'avg' is util.AvgMetric (util.py:144-150)
                value = (auc + mrr + (ndcg5 + ndcg10) / 2) / 3 if self.dev_criterion == 'avg' else dict(zip(_CRITERIA, (auc, mrr, ndcg5, ndcg10)))[self.dev_criterion]
                if value >= self.best:                                       # trainer.py:163-211
                    self.best, self.best_dev_epoch, self.epoch_not_increase = value, e, 0
                    with open(os.path.join(self.result_dir, '#%d-dev' % self.run_index), 'w') as f:
                        f.write('#%d\t%s\t%s\t%s\t%s\n' % (self.run_index, auc, mrr, ndcg5, ndcg10))
                    save_checkpoint(os.path.join(self.model_dir, '%s-%d' % (name, e)), self.model, self.step)   # trainer.py:220
                else:
                    self.epoch_not_increase += 1
            if self.world > 1:                                               # every rank follows rank 0's stopping decision
                flag = torch.tensor([self.epoch_not_increase, self.best_dev_epoch], device=next(self.model.parameters()).device)
                torch.distributed.broadcast(flag, 0)
                self.epoch_not_increase, self.best_dev_epoch = int(flag[0].item()), int(flag[1].item())
            if self.epoch_not_increase == self.early_stopping_epoch:
                break
        if self.rank == 0:
            with open(os.path.join(self.dev_res_dir, '%s-%s-dev_log.txt' % (name, self.config.dataset)), 'w', encoding='utf-8') as f:
                f.write('Epoch\tAUC\tMRR\tnDCG@5\tnDCG@10\n')
                for i in range(len(self.results['auc'])):
                    f.write('%d\t%.4f\t%.4f\t%.4f\t%.4f\n' % ((i + 1,) + tuple(self.results[k][i] for k in _CRITERIA)))
            shutil.copy(os.path.join(self.model_dir, '%s-%d' % (name, self.best_dev_epoch)), os.path.join(self.best_model_dir, name))
        return self.best_dev_epoch
