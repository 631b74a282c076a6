import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from lime_cikm25_amd import Model, make_config, synth, newsEncoders, ops
from lime_cikm25_amd.training import negative_log_softmax
cfg = make_config(vocabulary_size=4000, max_history_num=20, max_title_length=32, max_abstract_length=64, batch_size=16)
model = Model(cfg); model.initialize(); synth.fill_state_dict(model, seed=33); model = model.cuda()
model.eval(); model.training = True
batch = [v.cuda() for v in synth.make_batch(cfg, 16, 5, seed=34).values()]
res = {}
for dedup in (True, False):
    for split in (False, True):
        newsEncoders.DEDUP = dedup
        ops.set_split_gemm(split)
        model.zero_grad(set_to_none=True)
        loss = negative_log_softmax(model(*batch)); loss.backward()
        res[(dedup, split)] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
def cmp(a, b, tag):
    (la, ga), (lb, gb) = res[a], res[b]
    rows = []
    for k in gb:
        x, y = ga[k].double(), gb[k].double()
        rows.append((float((x - y).abs().max()) / (float(y.abs().max()) + 1e-12), k, float(y.abs().max())))
    rows.sort(reverse=True)
    print(tag, 'loss', la, lb)
    for r in rows[:6]: print('   %.3e  %s  (max |g| %.3e)' % r)
cmp((True, False), (False, False), 'fp32: dedup vs dense')
cmp((True, True), (False, True), 'split: dedup vs dense')
cmp((False, True), (False, False), 'dense: split vs fp32')
cmp((True, True), (True, False), 'dedup: split vs fp32')
