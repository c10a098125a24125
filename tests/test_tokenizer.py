"""ClipTokenizer against the reference's own class (tokenizer/clip.py:10-78) run on a toy merge table
(tests/golden/clip_bpe_toy.txt.gz + clip_tokens.json, produced by tests/golden/make_golden.py ``tokenizer``)."""
import gzip
import json
import os

import pytest

from tinyfusers_amd.tokenizer.clip import ClipTokenizer, byte_symbols

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tok():
    return ClipTokenizer(os.path.join(G, "clip_bpe_toy.txt.gz"))


def test_ids_match_the_reference_tokenizer(tok):
    g = json.load(open(os.path.join(G, "clip_tokens.json")))
    assert len(tok.encoder) == g["vocab_size"]
    for text, want in zip(g["texts"], g["ids"]):
        got = tok.encode(text)
        assert got == want, (text, got[:12], want[:12])
        assert len(got) == 77 and got[0] == 49406 and got[-1] == 49407


def test_byte_table_is_the_published_one():
    table, order = byte_symbols()
    assert len(table) == 256 and len(set(table.values())) == 256
    assert table[ord("a")] == "a" and table[ord(" ")] == chr(256 + 32) and table[0] == chr(256) and table[0xAD] == chr(256 + 67)
    assert order[:3] == ["!", '"', "#"] and order[188] == chr(256)


def test_truncation_padding_and_plain_text_file(tok, tmp_path):
    long = " ".join(["cat"] * 200)
    ids = tok.encode(long)
    assert len(ids) == 77 and ids[-1] == 49407 and 49407 not in ids[1:76]
    raw = gzip.open(os.path.join(G, "clip_bpe_toy.txt.gz")).read()
    p = os.path.join(tmp_path, "merges.txt")
    open(p, "wb").write(raw)
    assert ClipTokenizer(p).encode("a horse sized cat") == tok.encode("a horse sized cat")
    with pytest.raises(ValueError):
        ClipTokenizer("")
