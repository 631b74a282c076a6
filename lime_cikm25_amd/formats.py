"""On-disk formats of the reference: the lifetime-augmented ``behaviors.tsv`` and ``news.tsv``
lines, turned into the in-memory records ``device_data.DeviceBehaviors`` consumes -- the lists the reference's ``Corpus``
builds at corpus.py:478-552 (train) and :556-650 (dev / test) -- and the per-news arrays of corpus.py:360-367, :407-477 from the
news lines and the vocabulary / category dictionaries the reference's preprocessing wrote.  Host-side text work; building the
vocabulary, the GloVe table and the knowledge-graph files (corpus.py:27-296) stays with the reference's preprocessing.

Parity: pinned by tests/golden/formats.json -- the records the IMPORTED reference's ``Corpus`` parsed out of a synthetic dataset
directory (tools/make_format_goldens.py writes the tsv files in the reference's layout, runs corpus.py on them and stores its
train / dev / test records next to the lines); tests/test_formats.py requires equality, record for record.

Upstream defect kept visible: the dev / test loops look the candidate's topic up with ``news_idx`` -- the loop variable LEFT
OVER from the train loop (corpus.py:582, :630) -- instead of the candidate's own ``news_index``, so every dev / test candidate
gets the user-topic lifetime of the last news of the last train impression.  ``devtest_records`` does the lookup per candidate
by default and reproduces the reference when ``stale_news_index`` is given.
"""
import ast
import json
import os
import re

import numpy as np

NEWS_COLUMNS = ('news_ID', 'category', 'subCategory', 'title', 'abstract', 'publishTime', 'title_entities', 'abstract_entities')


def parse_news_line(line, strip=True):
    """One ``news.tsv`` line -> dict of its 8 columns.  The train split is read with ``line.split('\\t')`` (corpus.py:384: the last
    column keeps its newline), the MIND dev / test splits with ``line.strip().split('\\t', 7)`` (:391, :400): ``strip`` selects."""
    parts = line.strip().split('\t', 7) if strip else line.split('\t')
    if len(parts) != 8:
        raise ValueError('news.tsv line has %d columns, expected 8' % len(parts))
    return dict(zip(NEWS_COLUMNS, parts))


def parse_behavior_line(line):
    """One 6-column ``behaviors.tsv`` line (corpus.py:481-490, :558-566):
    impression_ID, user_ID, time_str, history_str, impressions_str, lifetime_str, with
      time_str      a Python literal [[clicked-news freshness ...], [clicked-news user-topic lifetimes ...], [candidate freshness]]
      lifetime_str  JSON [ {topic: lifetime} seen by the user, {topic: lifetime} unseen fallback, default lifetime ].
    """
    parts = line.split('\t')
    if len(parts) != 6:
        raise ValueError('behaviors.tsv line has %d columns, expected 6' % len(parts))
    impression_ID, user_ID, time_str, history_str, impressions_str, lifetime_str = parts
    lifetime_list = ast.literal_eval(time_str)
    category_lifetime, unseen, default = json.loads(lifetime_str)
    history = history_str.strip().split() if history_str.strip() else []
    impressions = []
    for imp in impressions_str.strip().split(' '):
        news_id, label = imp[:-2], imp[-1]                     # "<news id>-<0|1>"  (:499, :575)
        impressions.append((news_id, label))
    return {'impression_ID': impression_ID, 'user_ID': user_ID, 'freshness_list': lifetime_list[0],
            'user_topic_lifetime_list': lifetime_list[1], 'candidate_freshness_list': lifetime_list[2], 'history': history,
            'impressions': impressions, 'category_lifetime': category_lifetime, 'unseen': unseen, 'default_lifetime': default}


def topic_lifetime(topic, rec):
    """Seen-by-the-user lifetime, else the unseen-topic fallback, else the default (corpus.py:521-536, :583-588)."""
    if topic in rec['category_lifetime']:
        return rec['category_lifetime'][topic]
    if topic in rec['unseen']:
        return rec['unseen'][topic]
    return rec['default_lifetime']


def history_slots(history_ids, news_ID_dict, max_history_num):
    """The LAST max_history_num clicked news, zero-padded at the end, and the mask of real slots (corpus.py:510-518)."""
    history = [news_ID_dict[x] for x in history_ids]
    padded = history[-max_history_num:] + [0] * max(0, max_history_num - len(history))
    mask = np.zeros([max_history_num], dtype=bool)
    mask[:min(len(history), max_history_num)] = 1
    return padded, mask


def train_records(lines, news_ID_dict, user_ID_dict, news_category, category_index_to_name, max_history_num):
    """``Corpus.train_behaviors`` (corpus.py:478-552): one 11-field record per CLICKED impression of every line.  Returns the
    records and the value the reference's loop variable ``news_idx`` is left with (see the module docstring)."""
    out, news_idx = [], None
    for behavior_index, line in enumerate(lines):
        rec = parse_behavior_line(line)
        clicks, non_clicks, pos_topic, neg_topics = [], [], None, []
        for news_id, label in rec['impressions']:
            news_idx = news_ID_dict[news_id]
            topic = category_index_to_name[news_category[news_idx]]
            if label == '1':
                clicks.append(news_idx)
                pos_topic = topic
            else:
                non_clicks.append(news_idx)
                neg_topics.append(topic)
        user_history, mask = history_slots(rec['history'], news_ID_dict, max_history_num)
        pos_lifetime = topic_lifetime(pos_topic, rec)
        neg_lifetimes = [topic_lifetime(t, rec) for t in neg_topics]
        for click in clicks:
            out.append([user_ID_dict[rec['user_ID']], user_history, mask, click, non_clicks, behavior_index,
                        rec['candidate_freshness_list'][0], pos_lifetime, neg_lifetimes, rec['freshness_list'],
                        rec['user_topic_lifetime_list']])
    return out, news_idx


def devtest_records(lines, news_ID_dict, user_ID_dict, news_category, category_index_to_name, max_history_num,
                    stale_news_index=None):
    """``Corpus.dev_behaviors`` / ``test_behaviors`` and ``dev_indices`` / ``test_indices`` (corpus.py:556-650): one 9-field
    record per (impression, candidate).  ``stale_news_index``: reproduce the reference's topic lookup (module docstring)."""
    records, indices = [], []
    for row, line in enumerate(lines):
        rec = parse_behavior_line(line)
        user_history, mask = history_slots(rec['history'], news_ID_dict, max_history_num)
        for news_id, _label in rec['impressions']:
            news_index = news_ID_dict[news_id]
            lookup = news_index if stale_news_index is None else stale_news_index
            lifetime = topic_lifetime(category_index_to_name[news_category[lookup]], rec)
            indices.append(row)
            records.append([user_ID_dict.get(rec['user_ID'], 0), user_history, mask, news_index, row, rec['candidate_freshness_list'][0],
                            lifetime, rec['freshness_list'], rec['user_topic_lifetime_list']])
    return records, indices


def truth_labels(lines):
    """Per impression the 0 / 1 click labels in candidate order: what the truth file of config.py:262-276 holds."""
    return [[int(label) for _nid, label in parse_behavior_line(line)['impressions']] for line in lines]


def write_truth_file(path, labels):
    """The truth file of config.py:262-276: ``<impression> [l1,l2,...]`` lines, 1-based ids, no spaces, no trailing newline."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w', encoding='utf-8') as f:
        for i, lab in enumerate(labels):
            f.write(('' if i == 0 else '\n') + str(i + 1) + ' ' + str([int(v) for v in lab]).replace(' ', ''))
    return path


# ---- news.tsv -> the per-news arrays of corpus.py:360-367 (what DeviceCorpus keeps in HBM) ----------------------------------
_MIND_TOKENS = re.compile(r"[\w]+|[.,!?;|]")            # the 'MIND' tokenizer (corpus.py:23); 'NLTK' needs nltk and is not offered


def _is_number(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def _encode(text, length, word_dict, ids, mask):
    for i, word in enumerate(_MIND_TOKENS.findall(text.lower())):
        if i == length:
            break
        ids[i] = word_dict['<NUM>'] if _is_number(word) else word_dict.get(word, 1)      # 1 = <UNK> (corpus.py:421-426)
        mask[i] = True


def news_arrays(news_lines_by_split, news_ID_dict, category_dict, subCategory_dict, word_dict, max_title_length, max_abstract_length,
                dataset='mind'):
    """``Corpus.news_category / news_subCategory / news_title_text / news_title_mask / news_abstract_text / news_abstract_mask``
    (corpus.py:360-367, :380-477) from the news.tsv lines of the train, dev and test splits (in that order; a news keeps its
    FIRST line), the 'MIND' regex tokenizer and the dictionaries of the reference's preprocessing.  Row 0 is the <PAD> news
    (its masks have position 0 set, :476-477).  The entity arrays (unused by LIME-{CROWN,MHSA}-CROWN) are not produced."""
    n = len(news_ID_dict)
    out = {'news_category': np.zeros(n, np.int32), 'news_subCategory': np.zeros(n, np.int32),
           'news_title_text': np.zeros((n, max_title_length), np.int32), 'news_title_mask': np.zeros((n, max_title_length), bool),
           'news_abstract_text': np.zeros((n, max_abstract_length), np.int32),
           'news_abstract_mask': np.zeros((n, max_abstract_length), bool)}
    seen = {'<PAD>'}
    for split, lines in enumerate(news_lines_by_split):
        for line in lines:
            rec = parse_news_line(line, strip=(split > 0 and dataset == 'mind'))
            if rec['news_ID'] in seen:
                continue
            seen.add(rec['news_ID'])
            index = news_ID_dict[rec['news_ID']]
            out['news_category'][index] = category_dict.get(rec['category'], 0)
            out['news_subCategory'][index] = subCategory_dict.get(rec['subCategory'], 0)
            _encode(rec['title'], max_title_length, word_dict, out['news_title_text'][index], out['news_title_mask'][index])
            _encode(rec['abstract'], max_abstract_length, word_dict, out['news_abstract_text'][index], out['news_abstract_mask'][index])
    if len(seen) != n:
        raise ValueError('news_ID_dict has %d entries, the files hold %d news' % (n, len(seen)))
    out['news_title_mask'][0][0] = True
    out['news_abstract_mask'][0][0] = True
    return out


def build_corpus(config, news_lines_by_split, behavior_lines_by_split, news_ID_dict, user_ID_dict, category_dict, subCategory_dict,
                 word_dict, dataset='mind', reference_stale_lookup=False):
    """The ``Corpus`` attributes ``DeviceCorpus`` / ``DeviceBehaviors`` (and the reference's datasets) read, from the raw tsv
    lines of the (train, dev, test) splits and the dictionaries of the reference's preprocessing (its ``*-<dataset>.json`` files).
    ``reference_stale_lookup``: reproduce the dev / test topic lookup of corpus.py:582,:630 (module docstring)."""
    from types import SimpleNamespace
    c = SimpleNamespace(config=config, max_history_num=config.max_history_num, max_title_length=config.max_title_length,
                        max_abstract_length=config.max_abstract_length, negative_sample_num=config.negative_sample_num,
                        news_ID_dict=news_ID_dict, user_ID_dict=user_ID_dict, category_dict=category_dict,
                        subCategory_dict=subCategory_dict, word_dict=word_dict)
    for k, v in news_arrays(news_lines_by_split, news_ID_dict, category_dict, subCategory_dict, word_dict, config.max_title_length,
                            config.max_abstract_length, dataset).items():
        setattr(c, k, v)
    c.news_title_entity = np.zeros_like(c.news_title_text)                  # unused by LIME-{CROWN,MHSA}-CROWN
    c.news_abstract_entity = np.zeros_like(c.news_abstract_text)
    c.category_index_to_name = {v: k for k, v in category_dict.items()}
    args = (news_ID_dict, user_ID_dict, c.news_category, c.category_index_to_name, config.max_history_num)
    c.train_behaviors, left_over = train_records(behavior_lines_by_split[0], *args)
    stale = left_over if reference_stale_lookup else None
    c.dev_behaviors, c.dev_indices = devtest_records(behavior_lines_by_split[1], *args, stale_news_index=stale)
    c.test_behaviors, c.test_indices = devtest_records(behavior_lines_by_split[2], *args, stale_news_index=stale)
    c.dev_labels = truth_labels(behavior_lines_by_split[1])                 # what config.py:262-276 writes to dev/ref/truth-*.txt
    c.test_labels = truth_labels(behavior_lines_by_split[2])
    for split in ('train', 'dev', 'test'):                                  # SUE-only tables (dataset.py:21-28)
        for name in ('user_history_graph', 'user_history_category_mask', 'user_history_category_indices'):
            setattr(c, '%s_%s' % (split, name), None)
    return c
