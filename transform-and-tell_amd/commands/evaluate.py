"""tell/commands/evaluate.py:31-223 on the MI355X path: the test-set loop that generates a caption per sample (greedy or
beam, K/V-cached static-batch generator), accumulates the model's metrics and appends one JSON object per sample to
`generations<suffix>.jsonl`.

The reference annotates every record with spaCy (named entities / proper nouns), textstat (readability) and nltk
(type-token ratios).  Those are host-side text libraries that are not part of the GPU path and are absent here: they are
optional plug-ins (`annotate=`); without one, the record carries the fields that need no NLP model - the keys the
reference writes first (`caption`, `raw_caption`, `generation`, `copied_texts`, `web_url`, `image_path`, `context`) plus
the type-token ratios, which only need a tokeniser (whitespace + punctuation strip stands in for nltk.word_tokenize)."""
import json
import math
import os
import string

import torch

from .. import config as cfg


def _ttr(text):
    """evaluate.py:273-356 `get_narrative_productivity` (basic / root / corrected TTR, Herdan, Summer, Maas)."""
    words = [w.strip(string.punctuation) for w in text.split()]
    words = [w for w in words if w]
    w, t = len(words), len(set(words))
    lw, lt = (math.log(w) if w > 0 else 0.0), (math.log(t) if t > 0 else 0.0)
    return {
        'basic_ttr': t / w if w else 0,
        'root_ttr': t / math.sqrt(w) if w else 0,
        'corrected_ttr': t / math.sqrt(2 * w) if w else 0,
        'herdan': lt / lw if w > 1 else 0,
        'summer': math.log(lt) / math.log(lw) if (w > 2 and t > 2 and lt > 0 and lw > 1) else 0,
        'maas': (lw - lt) / (lw ** 2) if w > 1 else 0,
    }


def write_to_json(output_dict, serialization_dir, eval_suffix='', annotate=None):
    """evaluate.py:179-223.  annotate(record, metadata) may add the NLP-derived keys (caption_names, ...)."""
    if 'captions' not in output_dict:
        return
    captions, generations, metadatas = output_dict['captions'], output_dict['generations'], output_dict['metadata']
    copied = output_dict.get('copied_texts', [''] * len(captions))
    out_path = os.path.join(serialization_dir, 'generations%s.jsonl' % eval_suffix)
    with open(out_path, 'a') as f:
        for i, caption in enumerate(captions):
            m = metadatas[i]
            obj = {'caption': caption, 'raw_caption': m.get('caption'), 'generation': generations[i],
                   'copied_texts': copied[i], 'web_url': m.get('web_url'), 'image_path': m.get('image_path'),
                   'context': m.get('context'), 'caption_np': _ttr(m.get('caption') or ''),
                   'gen_np': _ttr(generations[i])}
            if annotate is not None:
                annotate(obj, m)
            f.write(json.dumps(obj) + '\n')


def evaluate(model, instances, data_iterator, cuda_device, serialization_dir, eval_suffix='', batch_weight_key='',
             annotate=None, beam_size=1):
    """evaluate.py:89-176: -> final metrics dict (model metrics + the weighted average loss)."""
    os.makedirs(serialization_dir, exist_ok=True)
    assert not os.path.exists(os.path.join(serialization_dir, 'generations%s.jsonl' % eval_suffix))
    device = torch.device(cuda_device) if not isinstance(cuda_device, int) else \
        torch.device('cuda', cuda_device) if cuda_device >= 0 else torch.device('cpu')
    with torch.no_grad():
        model.eval()
        model.evaluate_mode = True
        model.eval_beam_size = beam_size
        loss_count, total_loss, total_weight = 0, 0.0, 0.0
        batches = data_iterator(instances, num_epochs=1, shuffle=False, device=device)
        lanes = int(os.environ.get('TELL_EVAL_LANES', '2'))
        if hasattr(model, 'generate_lanes') and lanes > 1:
            # two batches' decode loops in flight together on two streams (CaptionModel.generate_lanes): +17-27 % captions/s
            outputs = (out for _, out in model.generate_lanes(batches, lanes=lanes, forward=True))
        elif hasattr(model, 'generate_stream'):       # encoders of batch N+1 underneath the decode loop of batch N
            outputs = (out for _, out in model.generate_stream(batches, forward=True))
        else:
            outputs = (model(**batch) for batch in batches)
        for output_dict in outputs:
            loss = output_dict.get('loss')
            write_to_json(output_dict, serialization_dir, eval_suffix, annotate)
            if loss is not None:
                loss_count += 1
                weight = float(output_dict[batch_weight_key]) if batch_weight_key else 1.0
                total_weight += weight
                total_loss += float(loss) * weight
        final_metrics = model.get_metrics(reset=True)
        if loss_count > 0:
            final_metrics['loss'] = total_loss / total_weight
    return final_metrics


def evaluate_from_file(archive_path, model_path=None, overrides='', eval_suffix='', shard_dir=None, device=None,
                       beam_size=1, **model_extra):
    """evaluate.py:31-86 for a `config.yaml`: build reader / vocabulary / model / validation iterator from the config,
    load `best.th` (a plain state_dict, :61-63), evaluate the test split, write `evaluate-metrics<suffix>.json`."""
    assert archive_path.endswith('yaml'), 'model archives (.tar.gz) are an AllenNLP format; pass the config.yaml'
    params = cfg.yaml_to_params(archive_path, overrides)
    serialization_dir = os.path.join(os.path.dirname(archive_path), 'serialization')
    reader = cfg.reader_from_params(params['dataset_reader'], **({'shard_dir': shard_dir} if shard_dir else {}))
    vocab = cfg.vocabulary_from_params(params['vocabulary'])
    model = cfg.model_from_params(params['model'], vocab, **model_extra)
    if model_path:
        model.load_state_dict(torch.load(model_path, map_location='cpu'))
    iterator = cfg.iterator_from_params(params['validation_iterator'])
    device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
    model.to(device)
    metrics = evaluate(model, reader._read(params.get('test_data_path', 'test')), iterator, device, serialization_dir,
                       eval_suffix, beam_size=beam_size)
    with open(os.path.join(serialization_dir, 'evaluate-metrics%s.json' % eval_suffix), 'w') as f:
        json.dump(metrics, f, indent=4)
    return metrics
