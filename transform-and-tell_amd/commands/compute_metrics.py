"""scripts/compute_metrics.py:61-298 - the caption metrics over a `generations.jsonl` written by the evaluation tail
(SURVEY 8-f4): BLEU-1..4, ROUGE-L, CIDEr, lengths / unique words, and - when the records carry the NLP-derived keys the
reference's evaluate writes (spaCy names and entities, textstat readability: an optional host-side plug-in here) - the
name / entity recall and precision tables, TTR and reading ease.  METEOR needs the reference's Java jar
(pycocoevalcap.meteor spawns `java -jar meteor-1.5.jar`): reported as None when no `meteor` callable is supplied."""
import json
import re
from collections import defaultdict

from ..metrics import BleuScorer
from ..metrics.cider import CiderScorer
from ..metrics.rouge import Rouge


def _ratio(count, total):
    return {'count': count, 'total': total, 'percentage': (count / total) if total else None}


def _names(obj, rare_counter=None):
    """(recall hits, recall total, precision hits, precision total) of caption_names vs generated_names; with a
    counter only the names it does NOT contain count (the 'rare names' tables, :375-395)."""
    cap, gen = obj.get('caption_names') or [], obj.get('generated_names') or []
    if rare_counter is not None:
        rcap = [n for n in cap if n not in rare_counter]
        rgen = [n for n in gen if n not in rare_counter]
    else:
        rcap, rgen = cap, gen
    return (sum(1 for n in rcap if n in gen), len(rcap), sum(1 for n in rgen if n in cap), len(rgen))


def _contains(entities, target):
    return any(e['text'] == target['text'] and e['label'] == target['label'] for e in entities)


def _entities(obj, c):
    """:262-350: matches by (text, label) between caption and generated entities, overall and per label."""
    cap, gen = obj.get('caption_entities'), obj.get('generated_entities')
    if cap is None or gen is None:
        return
    for label, key in ((None, 'ent'), ('PERSON', 'person'), ('ORG', 'orgs'), ('GPE', 'gpes'), ('DATE', 'date')):
        ce = [e for e in cap if label is None or e['label'] == label]
        ge = [e for e in gen if label is None or e['label'] == label]
        c['n_caption_' + key] += len(ce)
        c['n_gen_' + key] += len(ge)
        c['n_gen_%s_matches' % key] += sum(1 for e in ge if _contains(ce, e))
        c['n_caption_%s_matches' % key] += sum(1 for e in ce if _contains(ge, e))


def compute_metrics(path, counters=None, use_processed=False, meteor=None):
    """path: generations.jsonl.  counters: optional {'caption': Counter, 'context': Counter} of name frequencies in the
    training set (the reference's pickled counters, :71-74).  meteor: optional callable(generations, captions) -> float.
    -> dict with the reference's metric names (:179-298)."""
    bleu, rouge, cider = BleuScorer(n=4), Rouge(), CiderScorer(n=4, sigma=6.0)
    rouge_scores, lengths, gt_lengths, uniq, gt_uniq = [], [], [], [], []
    gens, caps = [], []
    tallies = defaultdict(lambda: [0, 0, 0, 0])
    ent = defaultdict(int)
    extras = defaultdict(list)
    cap_counter = counters['caption'] if counters else None
    full_counter = (counters['context'] + counters['caption']) if counters else None
    with open(path) as f:
        for line in f:
            obj = json.loads(line)
            if use_processed:
                caption = obj['caption']
                obj['caption_names'] = obj.get('processed_caption_names')
            else:
                caption = obj['raw_caption']
            generation = obj['generation']
            for key, ctr in (('all', None), ('caption_rare', cap_counter), ('article_rare', full_counter)):
                if key == 'all' or ctr is not None:
                    for i, v in enumerate(_names(obj, ctr)):
                        tallies[key][i] += v
            caption = re.sub(r'[^\w\s]', '', caption)                      # :139-141 remove punctuation
            generation = re.sub(r'[^\w\s]', '', generation)
            lengths.append(len(generation.split()))
            gt_lengths.append(len(caption.split()))
            uniq.append(len(set(generation.split())))
            gt_uniq.append(len(set(caption.split())))
            bleu += (generation, [caption])
            rouge_scores.append(rouge.calc_score([generation], [caption]))
            cider += (generation, [caption])
            gens.append(generation)
            caps.append(caption)
            for key, src, field in (('Caption TTR', 'caption_np', 'basic_ttr'), ('Generation TTR', 'gen_np', 'basic_ttr'),
                                    ('Caption Flesch Reading Ease', 'caption_readability', 'flesch_reading_ease'),
                                    ('Generation Flesch Reading Ease', 'gen_readability', 'flesch_reading_ease')):
                if isinstance(obj.get(src), dict) and field in obj[src]:
                    extras[key].append(obj[src][field])
            _entities(obj, ent)
    n = max(len(lengths), 1)
    b, _ = bleu.compute_score(option='closest')
    c, _ = cider.compute_score()
    out = {'BLEU-1': b[0], 'BLEU-2': b[1], 'BLEU-3': b[2], 'BLEU-4': b[3],
           'ROUGE': sum(rouge_scores) / n, 'METEOR': meteor(gens, caps) if meteor else None, 'CIDEr': c,
           'All names - recall': _ratio(*tallies['all'][:2]), 'All names - precision': _ratio(*tallies['all'][2:]),
           'Length - generation': sum(lengths) / n, 'Length - reference': sum(gt_lengths) / n,
           'Unique words - generation': sum(uniq) / n, 'Unique words - reference': sum(gt_uniq) / n}
    if counters:
        out['Caption rare names - recall'] = _ratio(*tallies['caption_rare'][:2])
        out['Caption rare names - precision'] = _ratio(*tallies['caption_rare'][2:])
        out['Article rare names - recall'] = _ratio(*tallies['article_rare'][:2])
        out['Article rare names - precision'] = _ratio(*tallies['article_rare'][2:])
    for key, vals in extras.items():
        out[key] = sum(vals) / len(vals)
    for label, key in (('all', 'ent'), ('person', 'person'), ('GPE', 'gpes'), ('ORG', 'orgs'), ('DATE', 'date')):
        if ent['n_caption_' + key] or ent['n_gen_' + key]:
            out['Entity %s - recall' % label] = _ratio(ent['n_caption_%s_matches' % key], ent['n_caption_' + key])
            out['Entity %s - precision' % label] = _ratio(ent['n_gen_%s_matches' % key], ent['n_gen_' + key])
    return out
