"""Task metrics of the evaluation harness, the part that needs neither COCO nor the pycocoevalcap / detection_metrics packages
(offline remainder): the scoring rules of exp/gpv/evaluators.py:32-127 and exp/gpv/metrics.py:15-66.

  * CocoVqa.evaluate              (evaluators.py:36-82)    soft VQA accuracy min(#annotators / 3, 1) on the lower-cased answer, broken
                                                            down by answer_type / question_type, percentages rounded to 2 places
  * CocoClassification.evaluate   (evaluators.py:89-127)   correct iff the lower-cased prediction is one of the class's synonyms
  * vqa_accuracy_from_predictions (metrics.py:49-65)       the train-time variant: NO lower-casing, fraction rounded to 4 places

Captioning (Bleu / Cider) and detection / referring-expression mAP call third-party scorers the reference keeps under third_party/
(empty in the checkout: un-vendored); their inputs are the prediction files gpv1_amd.compute_predictions writes in the reference's
layout, so the reference's evaluators consume them as they are.

Pinned by tests/golden/evaluators.json, produced by the reference's own classes (tools/gen_golden_evaluators.py)."""
from collections import Counter

TASK_TO_ID = {'CocoVqa': 'question_id', 'CocoClassification': 'id', 'CocoCaptioning': 'cap_id', 'CocoDetection': 'id', 'RefCocop': 'sent_id'}
EPS = 1e-6


class CocoEval:
    """evaluators.py:17-29: samples keyed by str(sample[<task id>]); predictions = {key: {'answer': str}}"""

    def __init__(self, samples, predictions, boxes, task):
        self.task = task
        self.task_id_name = TASK_TO_ID[task]
        self.samples = {str(s[self.task_id_name]): s for s in samples}
        self.predictions = predictions
        self.boxes = boxes

    def sample_novelty(self, sample):
        return 'held_out_concepts' if len(sample['coco_categories']['unseen']) > 0 else 'seen_concepts'

    def _selected(self, novelty):
        """-> (key, sample) of the samples that count, and the number of selected samples without a prediction"""
        absent, picked = 0, []
        for k, sample in self.samples.items():
            if novelty != 'everything' and self.sample_novelty(sample) != novelty:
                continue
            if k not in self.predictions:
                absent += 1
                continue
            picked.append((k, sample))
        return picked, absent


class CocoVqa(CocoEval):
    def __init__(self, samples, predictions, boxes=None, task='CocoVqa'):
        super().__init__(samples, predictions, boxes, task)

    def evaluate(self, novelty='everything'):
        correct = {'all': 0, 'answer_type': Counter(), 'question_type': Counter()}
        total = {'all': 0, 'answer_type': Counter(), 'question_type': Counter()}
        picked, absent = self._selected(novelty)
        for k, sample in picked:
            pred = self.predictions[k]['answer'].lower()
            gt = {a.lower(): n for a, n in sample['all_answers'].items()}      # (later duplicates after lower-casing win, as in the reference)
            at, qt = sample['anno']['answer_type'], sample['anno']['question_type']
            if pred in gt:
                c = min(gt[pred] / 3, 1)
                correct['all'] += c
                correct['answer_type'][at] += c
                correct['question_type'][qt] += c
            total['all'] += 1
            total['answer_type'][at] += 1
            total['question_type'][qt] += 1
        accuracy = {'all': round(100 * correct['all'] / (EPS + total['all']), 2)}
        for key in ('answer_type', 'question_type'):
            accuracy[key] = {a: round(100 * correct[key][a] / (EPS + total[key][a]), 2) for a in total[key]}
        return {'correct': correct, 'total': total, 'absent': absent, 'accuracy': accuracy}


class CocoClassification(CocoEval):
    """synonyms: {coco class: [names]} -- the reference's data/coco/synonyms.py table (data the caller supplies; every class maps at
    least to itself there)"""

    def __init__(self, samples, predictions, boxes=None, task='CocoClassification', synonyms=None):
        super().__init__(samples, predictions, boxes, task)
        if synonyms is None:
            raise ValueError('CocoClassification needs the class -> synonyms table (data/coco/synonyms.py SYNONYMS)')
        self.synonyms = synonyms

    def evaluate(self, novelty='everything'):
        correct, total = Counter(), Counter()
        overall_correct = overall_total = 0
        picked, absent = self._selected(novelty)
        for k, sample in picked:
            pred = self.predictions[k]['answer'].lower()
            if pred in self.synonyms[sample['answer']]:
                overall_correct += 1
                correct[sample['answer']] += 1
            overall_total += 1
            total[sample['answer']] += 1
        return {'correct': correct, 'overall_correct': overall_correct, 'total': total, 'overall_total': overall_total, 'absent': absent,
                'accuracy': {k: round(correct[k] / (EPS + total[k]), 4) for k in total},
                'overall_accuracy': round(overall_correct / (EPS + overall_total), 4)}


def vqa_accuracy_from_predictions(pred_answers, samples, limit=None):
    """metrics.py:49-65 (the train-time VQA number): pred_answers[i] = the detokenised greedy answer of samples[i]; exact-case match
    against samples[i]['all_answers'] (no lower-casing here, unlike CocoVqa.evaluate), soft score min(n / 3, 1), at most `limit`
    samples; -> round(correct / (total + 1e-6), 4)"""
    correct, total = 0, 0
    for pred, sample in zip(pred_answers, samples):
        if limit is not None and total >= limit:
            break
        answers = sample['all_answers']
        if pred in answers:
            correct += min(answers[pred] / 3, 1)
        total += 1
    return round(correct / (total + 1e-6), 4)
