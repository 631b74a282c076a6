"""On-disk formats of the reference: the lifetime-augmented ``behaviors.tsv`` and ``news.tsv``
lines, turned into the in-memory records ``device_data.DeviceBehaviors`` consumes -- the lists the reference's ``Corpus``
builds at corpus.py:478-552 (train) and :556-650 (dev / test).  Host-side text parsing; no tokenisation (vocabulary building,
GloVe lookup and the knowledge-graph files of corpus.py:27-296 stay with the reference's preprocessing).

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
