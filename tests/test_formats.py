"""On-disk formats: the 6-column behaviors.tsv and 8-column news.tsv lines of the reference
(corpus.py:384-400, :478-650) through lime_cikm25_amd.formats: hand-written fixtures first (readable expectations), then
equality with what the imported reference's Corpus parsed out of a synthetic dataset (tests/golden/formats.json)."""
import json
from types import SimpleNamespace

import numpy as np
import pytest

from lime_cikm25_amd import formats, make_config
from oracle import lime_oracle as O

NEWS_ID = {'<PAD>': 0, 'N1': 1, 'N2': 2, 'N3': 3, 'N4': 4, 'N5': 5}
USER_ID = {'<UNK>': 0, 'U7': 1, 'U9': 2}
NEWS_CATEGORY = np.array([0, 1, 2, 1, 3, 2], dtype=np.int32)
CAT_NAME = {0: '<PAD>', 1: 'sports', 2: 'news', 3: 'finance'}


def behavior_line(imp, user, fresh, life, cand_fresh, history, impressions, seen, unseen, default):
    return '\t'.join([imp, user, repr([fresh, life, cand_fresh]), ' '.join(history), ' '.join(impressions),
                      json.dumps([seen, unseen, default])]) + '\n'


TRAIN = [
    behavior_line('1', 'U7', [100.0, 200.0], [3600.0, 7200.0], [50.0], ['N1', 'N2'], ['N3-1', 'N4-0', 'N5-0'],
                  {'sports': 1000.0}, {'finance': 2000.0}, 3000.0),
    behavior_line('2', 'U9', [], [], [75.0], [], ['N1-0', 'N2-1', 'N5-1'], {}, {'news': 500.0}, 9.0),
    behavior_line('3', 'U7', [1.0, 2.0, 3.0, 4.0], [5.0, 6.0, 7.0, 8.0], [60.0], ['N1', 'N2', 'N3', 'N4'], ['N4-1'],
                  {'finance': 11.0}, {}, 12.0),
]
DEV = [
    behavior_line('10', 'U404', [10.0], [20.0], [30.0], ['N5'], ['N1-0', 'N4-1'], {'sports': 1.5}, {'finance': 2.5}, 3.5),
    behavior_line('11', 'U9', [], [], [40.0], [], ['N2-1'], {}, {}, 4.5),
]


def test_news_line_columns():
    line = 'N1\tsports\tsoccer\tA title\tAn abstract\t2019-11-11\t[]\t[{"x": 1}]\n'
    rec = formats.parse_news_line(line)
    assert list(rec) == list(formats.NEWS_COLUMNS) and rec['abstract_entities'] == '[{"x": 1}]'
    assert formats.parse_news_line(line, strip=False)['abstract_entities'].endswith('\n')      # corpus.py:384 keeps the newline
    with pytest.raises(ValueError):
        formats.parse_news_line('N1\tsports\n')


def test_behavior_line_fields():
    rec = formats.parse_behavior_line(TRAIN[0])
    assert rec['impression_ID'] == '1' and rec['user_ID'] == 'U7' and rec['history'] == ['N1', 'N2']
    assert rec['impressions'] == [('N3', '1'), ('N4', '0'), ('N5', '0')]
    assert rec['freshness_list'] == [100.0, 200.0] and rec['candidate_freshness_list'] == [50.0]
    assert rec['category_lifetime'] == {'sports': 1000.0} and rec['unseen'] == {'finance': 2000.0} and rec['default_lifetime'] == 3000.0
    assert formats.parse_behavior_line(TRAIN[1])['history'] == []
    with pytest.raises(ValueError):
        formats.parse_behavior_line('1\tU7\t[]\n')


def test_train_records_follow_corpus_py():
    recs, left_over = formats.train_records(TRAIN, NEWS_ID, USER_ID, NEWS_CATEGORY, CAT_NAME, max_history_num=3)
    assert len(recs) == 4                                            # one record per CLICKED impression: 1 + 2 + 1
    r = recs[0]
    assert r[0] == 1 and r[1] == [1, 2, 0] and r[2].tolist() == [True, True, False]
    assert r[3] == 3 and r[4] == [4, 5] and r[5] == 0 and r[6] == 50.0
    assert r[7] == 1000.0                                            # N3 is 'sports': seen by the user
    assert r[8] == [2000.0, 3000.0]                                  # N4 'finance' -> unseen table, N5 'news' -> default
    assert r[9] == [100.0, 200.0] and r[10] == [3600.0, 7200.0]
    # the second line has two clicks: both records share the line's negatives, and the positive lifetime is that of the LAST
    # clicked topic (pos_topic is overwritten, corpus.py:503-505)
    assert [x[3] for x in recs[1:3]] == [2, 5] and recs[1][4] == recs[2][4] == [1]
    assert recs[1][7] == recs[2][7] == 500.0 and recs[1][1] == [0, 0, 0] and not recs[1][2].any()
    # a history longer than max_history_num keeps its LAST entries
    assert recs[3][1] == [2, 3, 4] and recs[3][2].all()
    assert left_over == 4                                            # the value the reference's dev / test loops would reuse


def test_devtest_records_and_the_stale_index_defect():
    recs, idx = formats.devtest_records(DEV, NEWS_ID, USER_ID, NEWS_CATEGORY, CAT_NAME, max_history_num=3)
    assert idx == [0, 0, 1] and [r[3] for r in recs] == [1, 4, 2] and [r[4] for r in recs] == [0, 0, 1]
    assert recs[0][0] == 0                                           # unknown user -> 0 (corpus.py:591)
    assert [r[6] for r in recs] == [1.5, 2.5, 4.5]                   # per-candidate topic: sports, finance, default
    assert recs[0][5] == 30.0 and recs[0][7] == [10.0] and recs[0][8] == [20.0]
    # the reference looks every candidate's topic up with the index left over from the train loop (here 4 = 'finance')
    stale, _ = formats.devtest_records(DEV, NEWS_ID, USER_ID, NEWS_CATEGORY, CAT_NAME, 3, stale_news_index=4)
    assert [r[6] for r in stale] == [2.5, 2.5, 4.5]
    assert formats.truth_labels(DEV) == [[0, 1], [1]]


def test_records_feed_the_batch_assembly():
    """The records are what dataset.py consumes: through the oracle's assemble_* (pinned by the dataset goldens)."""
    cfg = make_config(max_history_num=3, max_title_length=4, max_abstract_length=5, vocabulary_size=50, negative_sample_num=2)
    n = len(NEWS_ID)
    c = SimpleNamespace(config=cfg, max_history_num=3, negative_sample_num=2, max_title_length=4, max_abstract_length=5,
                        news_category=NEWS_CATEGORY, news_subCategory=NEWS_CATEGORY.copy())
    for name, L in (('title', 4), ('abstract', 5)):
        setattr(c, 'news_%s_text' % name, np.arange(n * L, dtype=np.int32).reshape(n, L))
        setattr(c, 'news_%s_mask' % name, np.ones((n, L), dtype=bool))
        setattr(c, 'news_%s_entity' % name, np.zeros((n, L), dtype=np.int32))
    c.train_behaviors, _ = formats.train_records(TRAIN, NEWS_ID, USER_ID, NEWS_CATEGORY, CAT_NAME, 3)
    c.dev_behaviors, _ = formats.devtest_records(DEV, NEWS_ID, USER_ID, NEWS_CATEGORY, CAT_NAME, 3)
    c.test_behaviors = c.dev_behaviors
    out = O.assemble_devtest(c, 'dev', [0, 1, 2])
    assert out[0].tolist() == [0, 0, 2]                              # user ids
    assert np.asarray(out[15]).tolist() == [1, 3, 2]                 # news_category of candidates N1, N4, N2


# ---- pinned against the reference: tests/golden/formats.json holds what the imported reference's Corpus (corpus.py) parsed out
# ---- of a synthetic dataset directory (tools/make_format_goldens.py); the tsv lines are stored beside its records
def _golden():
    import os
    from helpers import GOLDEN_DIR
    return json.load(open(os.path.join(GOLDEN_DIR, 'formats.json')))


def _norm(records):
    return [[x.tolist() if isinstance(x, np.ndarray) else x for x in r] for r in records]


def test_train_records_equal_the_reference_corpus():
    g = _golden()
    names = {int(k): v for k, v in g['category_index_to_name'].items()}
    recs, left_over = formats.train_records(g['lines']['train_behaviors'], g['news_ID_dict'], g['user_ID_dict'], g['news_category'],
                                            names, g['max_history_num'])
    assert _norm(recs) == g['train_behaviors']
    assert left_over is not None


@pytest.mark.parametrize('split', ['dev', 'test'])
def test_devtest_records_equal_the_reference_corpus(split):
    """With the stale index the reference's loops reuse (corpus.py:582, :630) the records are the reference's, bit for bit; the
    per-candidate lookup (the default) differs exactly in the user-topic lifetime column."""
    g = _golden()
    names = {int(k): v for k, v in g['category_index_to_name'].items()}
    _, left_over = formats.train_records(g['lines']['train_behaviors'], g['news_ID_dict'], g['user_ID_dict'], g['news_category'],
                                         names, g['max_history_num'])
    args = (g['lines'][split + '_behaviors'], g['news_ID_dict'], g['user_ID_dict'], g['news_category'], names, g['max_history_num'])
    recs, idx = formats.devtest_records(*args, stale_news_index=left_over)
    assert _norm(recs) == g[split + '_behaviors'] and idx == g[split + '_indices']
    fixed, idx2 = formats.devtest_records(*args)
    assert idx2 == idx
    diff_cols = {j for a, b in zip(_norm(fixed), g[split + '_behaviors']) for j, (x, y) in enumerate(zip(a, b)) if x != y}
    assert diff_cols <= {6}


def test_news_lines_of_the_golden_dataset_parse():
    g = _golden()
    for split in ('train', 'dev', 'test'):
        for line in g['lines'][split + '_news']:
            rec = formats.parse_news_line(line, strip=split != 'train')
            assert rec['news_ID'] in g['news_ID_dict'] and rec['category'] in g['category_dict']


def test_news_arrays_equal_the_reference_corpus():
    """Token ids (vocabulary, <NUM>, <UNK>, truncation), masks and category ids as the reference's Corpus built them."""
    g = _golden()
    L = g['lines']
    got = formats.news_arrays([L['train_news'], L['dev_news'], L['test_news']], g['news_ID_dict'], g['category_dict'],
                              g['subCategory_dict'], g['word_dict'], g['max_title_length'], g['max_abstract_length'], dataset='adressa')
    for k in ('news_category', 'news_subCategory', 'news_title_text', 'news_abstract_text'):
        assert got[k].tolist() == g[k], k
    for k in ('news_title_mask', 'news_abstract_mask'):
        assert got[k].astype(int).tolist() == g[k], k
    words = {w for row in g['news_title_text'] + g['news_abstract_text'] for w in row}
    assert 1 in words and g['word_dict']['<NUM>'] in words and 0 in words          # <UNK>, <NUM> and padding all occur


def test_build_corpus_from_the_golden_files_feeds_the_batch_assembly():
    """Raw lines + the preprocessing's dictionaries -> a corpus object whose dev batch (through the oracle's assembly, pinned by
    the dataset goldens) carries the token rows and lifetimes of the reference's Corpus."""
    g = _golden()
    L = g['lines']
    cfg = make_config(max_history_num=g['max_history_num'], max_title_length=g['max_title_length'],
                      max_abstract_length=g['max_abstract_length'], vocabulary_size=len(g['word_dict']), negative_sample_num=2,
                      category_num=len(g['category_dict']) + 1)
    c = formats.build_corpus(cfg, [L['train_news'], L['dev_news'], L['test_news']],
                             [L['train_behaviors'], L['dev_behaviors'], L['test_behaviors']], g['news_ID_dict'], g['user_ID_dict'],
                             g['category_dict'], g['subCategory_dict'], g['word_dict'], dataset='adressa', reference_stale_lookup=True)
    assert _norm(c.dev_behaviors) == g['dev_behaviors'] and c.test_indices == g['test_indices']
    rows = list(range(len(c.dev_behaviors)))
    batch = O.assemble_devtest(c, 'dev', rows)
    cand = [r[3] for r in g['dev_behaviors']]
    assert np.asarray(batch[17]).tolist() == [g['news_title_text'][i] for i in cand]          # news_title_text of the candidates
    assert np.allclose(np.asarray(batch[24]), [r[6] for r in g['dev_behaviors']])              # news_user_topic_lifetime


def test_truth_file_is_the_reference_preprocessing_format(tmp_path):
    """config.py:262-276: '<impression> [l1,l2,...]' lines, 1-based, no spaces, no trailing newline; build_corpus attaches the
    labels the Trainer writes it from."""
    lines = _golden()['lines']['dev_behaviors']
    labels = formats.truth_labels(lines)
    path = formats.write_truth_file(str(tmp_path / 'dev' / 'ref' / 'truth-mind.txt'), labels)
    text = open(path).read()
    want = []
    for i, line in enumerate(lines):                                        # the reference's own loop body, restated
        impressions = line.split('\t')[4]
        lab = [int(imp[-1]) for imp in impressions.strip().split(' ')]
        want.append(('' if i == 0 else '\n') + str(i + 1) + ' ' + str(lab).replace(' ', ''))
    assert text == ''.join(want) and not text.endswith('\n')
    from lime_cikm25_amd.evaluate import parse_line
    assert [parse_line(l)[1] for l in text.split('\n')] == labels
