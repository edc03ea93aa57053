"""WordPiece tokenizer of the query encoder (gpv1_amd.bert.WordPieceTokenizer) against the reference's tokenizer call
(exp/gpv/models/bert.py:12-15: ``BertTokenizer(sentences, padding=True, return_tensors='pt')``) on a synthetic vocabulary in
bert-base-uncased's layout (tests/golden/bert_vocab_synthetic.txt -- the real vocab.txt is not available offline).  The
installed ``transformers`` BertTokenizer is the checker: ids and attention masks must be identical."""
import os

import pytest
import torch

VOCAB = os.path.join(os.path.dirname(__file__), 'golden', 'bert_vocab_synthetic.txt')

SENTENCES = [
    'What is this?',
    'what color is the umbrella',
    'Locate the man playing tennis.',
    'How many people are there?',
    "Don't run!  It's   a  dog,cat; or... a frisbee-skateboard",
    'Describe the image',
    'A naïve café résumé in New York — über Straße',          # accents stripped, ß kept, em dash is punctuation
    'unrunnable tokenization xyzzyq 12345 a1b2',
    'colour of the 中国 object',                               # CJK characters are split out as single tokens
    'tab\tand\nnewline\x00control\x7f chars�',
    '',
    '   ',
    'supercalifragilisticexpialidocious' * 4,                  # > 100 characters -> [UNK]
    '[CLS] hello [SEP] world [MASK] [UNK] [PAD]',              # special tokens in the text are never split
    "it’s “quoted”",
    'RUNNING Runner runs runnest',
]


def _hf():
    transformers = pytest.importorskip('transformers')
    return transformers.BertTokenizer(VOCAB, do_lower_case=True)


def test_wordpiece_matches_hf_bert_tokenizer_batch_padding():
    from gpv1_amd.bert import WordPieceTokenizer
    tok = WordPieceTokenizer(VOCAB)
    hf = _hf()
    ref = hf(SENTENCES, padding=True, return_tensors='pt')
    ids, attn = tok(SENTENCES)
    assert ids.dtype == torch.long and attn.dtype == torch.long
    assert ids.shape == ref['input_ids'].shape, (ids.shape, ref['input_ids'].shape)
    bad = [(s, ids[i].tolist(), ref['input_ids'][i].tolist()) for i, s in enumerate(SENTENCES) if not torch.equal(ids[i], ref['input_ids'][i])]
    assert not bad, bad[:3]
    assert torch.equal(attn, ref['attention_mask'])


@pytest.mark.parametrize('i', range(len(SENTENCES)))
def test_wordpiece_matches_hf_single_sentence(i):
    from gpv1_amd.bert import WordPieceTokenizer
    tok = WordPieceTokenizer(VOCAB)
    ref = _hf()([SENTENCES[i]], padding=True, return_tensors='pt')
    ids, attn = tok([SENTENCES[i]])
    assert torch.equal(ids, ref['input_ids']), (SENTENCES[i], ids.tolist(), ref['input_ids'].tolist())
    assert torch.equal(attn, ref['attention_mask'])


def test_bert_module_tokenises_strings_like_the_reference_call():
    """Bert.forward(list[str]) -> token_inputs with the reference's keys (bert.py:12-21); checked on the host part only"""
    from gpv1_amd.bert import Bert
    b = Bert(vocab_file=VOCAB, num_layers=1)
    assert b.tokenizer is not None
    ids, attn = b.tokenizer(['what is this', 'locate the dog playing frisbee'])
    ref = _hf()(['what is this', 'locate the dog playing frisbee'], padding=True, return_tensors='pt')
    assert torch.equal(ids, ref['input_ids']) and torch.equal(attn, ref['attention_mask'])
