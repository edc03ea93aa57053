"""gpv1_amd.evaluators against the fixture the reference's own CocoVqa / CocoClassification classes produced
(tools/gen_golden_evaluators.py, exp/gpv/evaluators.py:32-127, exp/gpv/metrics.py:54-65)."""
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'evaluators.json')


def _plain(x):
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    return x


def test_vqa_and_classification_scores_equal_the_reference():
    from gpv1_amd import evaluators as E
    g = json.load(open(GOLD))
    for nov, want in g['vqa']['expected'].items():
        got = _plain(E.CocoVqa(g['vqa']['samples'], g['vqa']['predictions']).evaluate(nov))
        assert got == want, nov
    for nov, want in g['cls']['expected'].items():
        got = _plain(E.CocoClassification(g['cls']['samples'], g['cls']['predictions'], synonyms=g['cls']['synonyms']).evaluate(nov))
        assert got == want, nov
    assert g['vqa']['expected']['everything']['absent'] > 0 and 0 < g['vqa']['expected']['everything']['accuracy']['all'] < 100
    assert 0 < g['cls']['expected']['everything']['overall_accuracy'] < 1


def test_train_time_vqa_accuracy_rule():
    from gpv1_amd import evaluators as E
    g = json.load(open(GOLD))
    preds = g['metrics_vqa_rule']['pred_answers']
    assert E.vqa_accuracy_from_predictions(preds, g['vqa']['samples']) == g['metrics_vqa_rule']['None']
    assert E.vqa_accuracy_from_predictions(preds, g['vqa']['samples'], limit=25) == g['metrics_vqa_rule']['25']
