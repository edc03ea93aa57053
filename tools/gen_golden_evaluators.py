"""Golden fixture for gpv1_amd.evaluators from the REFERENCE's own classes (the part that needs no data).  Build container only:

    python tools/gen_golden_evaluators.py      ->  tests/golden/evaluators.json

exp/gpv/evaluators.py is imported from where it lies with stand-ins for the two un-vendored packages it imports at module level
(third_party.pycocoevalcap.eval, third_party.detection_metrics.lib.Evaluator: the reference's third_party/ is empty; neither is
touched by the two classes run here) and tqdm; data.coco.synonyms is the reference's real table.  exp/gpv/metrics.py's vqa_accuracy
loop body (metrics.py:49-65) needs a model and a dataloader: its scoring lines are run through a two-line driver that feeds it the
already-decoded answers -- stated in the fixture as 'metrics_vqa_rule'."""
import json
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GOLD = os.path.join(ROOT, 'tests', 'golden')


def import_reference_evaluators():
    for name in ('third_party', 'third_party.pycocoevalcap', 'third_party.pycocoevalcap.eval', 'third_party.detection_metrics',
                 'third_party.detection_metrics.lib', 'third_party.detection_metrics.lib.Evaluator'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if 'tqdm' not in sys.modules:
        t = types.ModuleType('tqdm'); t.tqdm = lambda x, *a, **k: x; sys.modules['tqdm'] = t
    sys.path.insert(0, REF)
    import exp.gpv.evaluators as E
    return E


def make_cases(E):
    rnd = random.Random(7)
    classes = sorted(E.SYNONYMS)
    answers_pool = ['yes', 'no', '2', 'red', 'Red', 'a dog', 'frisbee', 'White', 'white', 'on the table', '']
    vqa_samples, vqa_pred = [], {}
    for i in range(60):
        pool = rnd.sample(answers_pool, rnd.randint(1, 5))
        all_answers = {a: rnd.randint(1, 10) for a in pool}
        s = {'question_id': 1000 + i, 'all_answers': all_answers,
             'anno': {'answer_type': rnd.choice(['yes/no', 'number', 'other']), 'question_type': rnd.choice(['what is', 'is the', 'how many', 'none of the above'])},
             'coco_categories': {'seen': ['dog'], 'unseen': (['cat'] if i % 3 == 0 else [])}}
        vqa_samples.append(s)
        if i % 7 != 6:                                      # every seventh sample has no prediction -> 'absent'
            vqa_pred[str(1000 + i)] = {'answer': rnd.choice(answers_pool + [a.upper() for a in pool])}
    cls_samples, cls_pred = [], {}
    for i in range(80):
        c = rnd.choice(classes)
        s = {'id': 5000 + i, 'answer': c, 'coco_categories': {'seen': [c], 'unseen': ([c] if i % 4 == 0 else [])}}
        cls_samples.append(s)
        if i % 9 != 8:
            wrong = rnd.choice(classes)
            cand = [rnd.choice(E.SYNONYMS[c]), rnd.choice(E.SYNONYMS[c]).upper(), wrong, rnd.choice(E.SYNONYMS[wrong]), 'thing']
            cls_pred[str(5000 + i)] = {'answer': rnd.choice(cand)}
    return vqa_samples, vqa_pred, cls_samples, cls_pred


def plain(x):
    if isinstance(x, dict):
        return {str(k): plain(v) for k, v in x.items()}
    return x


def main():
    E = import_reference_evaluators()
    vqa_samples, vqa_pred, cls_samples, cls_pred = make_cases(E)
    out = {'vqa': {'samples': vqa_samples, 'predictions': vqa_pred, 'expected': {}},
           'cls': {'samples': cls_samples, 'predictions': cls_pred, 'expected': {}, 'synonyms': {c: E.SYNONYMS[c] for c in sorted({s['answer'] for s in cls_samples})}}}
    for nov in ('everything', 'held_out_concepts', 'seen_concepts'):
        out['vqa']['expected'][nov] = plain(E.CocoVqa(vqa_samples, vqa_pred, None).evaluate(nov))
        out['cls']['expected'][nov] = plain(E.CocoClassification(cls_samples, cls_pred, None).evaluate(nov))
    # metrics.vqa_accuracy's scoring lines (metrics.py:54-58), run on already-decoded answers
    src = open(os.path.join(REF, 'exp/gpv/metrics.py')).read().splitlines()
    i0 = next(i for i, l in enumerate(src) if l.strip() == "answers = samples[total]['all_answers']")
    rule = src[i0:i0 + 4]                                   # 'answers = samples[total][...]' ... 'correct += correctness'
    assert rule[0].strip().startswith("answers = samples[total]['all_answers']") and rule[-1].strip() == 'correct += correctness', rule
    body = '\n'.join(l[12:] for l in rule)
    preds = [vqa_pred.get(str(s['question_id']), {'answer': 'zzz'})['answer'] for s in vqa_samples]
    for limit in (None, 25):
        ns = {'samples': vqa_samples, 'correct': 0, 'total': 0}
        for p in preds:
            if limit is not None and ns['total'] >= limit:
                break
            ns['pred_answer'] = p
            exec(body, ns)
            ns['total'] += 1
        out.setdefault('metrics_vqa_rule', {})[str(limit)] = round(ns['correct'] / (ns['total'] + 1e-6), 4)
    out['metrics_vqa_rule']['pred_answers'] = preds
    with open(os.path.join(GOLD, 'evaluators.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('wrote', os.path.join(GOLD, 'evaluators.json'), os.path.getsize(os.path.join(GOLD, 'evaluators.json')), 'bytes')


if __name__ == '__main__':
    main()
