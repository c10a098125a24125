"""ClipTokenizer -- mirrors tinyfusers/tokenizer/clip.py:10-78: lower-cased, whitespace-collapsed text ->
byte-level BPE ids -> [49406] + at most 75 ids + 49407 padding to 77 (example/sd1.py:44-48).

Pure host code.  The reference fetches ``bpe_simple_vocab_16e6.txt.gz`` from GitHub while its module is being imported
(:8, :11); there is no network here, so the merges file is an explicit local path.  Same file format and the same
vocabulary numbering: 256 byte symbols, the same 256 with the end-of-word mark, one id per merge line (lines
1 .. 48894 of the file), then the two specials.

The merge loop is organised differently from the reference's (which rebuilds the whole symbol tuple once per applied
merge and rescans it with ``tuple.index``): every adjacent pair's rank is looked up once, the best-ranked pair is merged
everywhere in one left-to-right pass, and only then are the ranks recomputed -- the greedy lowest-rank-first result is
the same by construction and is pinned against the reference's own class in tests/test_tokenizer.py."""
import gzip
import re

__all__ = ["ClipTokenizer", "byte_symbols"]

_END = "</w>"
_START_ID, _END_ID, _CONTEXT = 49406, 49407, 77          # tokenizer/clip.py:77-78
_N_MERGES = 49152 - 256 - 2                               # tokenizer/clip.py:15
# tokenizer/clip.py:25: the specials, the English clitics, otherwise any run of non-space characters
_WORD = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\s]+", re.IGNORECASE)


def byte_symbols():
    """byte value -> printable stand-in character (the GPT-2 / CLIP byte table, tokenizer/clip.py:85-103): bytes that
    are printable and not whitespace in Latin-1 stand for themselves, the other 68 get code points from 256 upwards."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, spare = {}, 256
    for b in sorted(keep):
        table[b] = chr(b)
    for b in range(256):
        if b not in keep:
            table[b] = chr(spare)
            spare += 1
    # numbering order of the vocabulary: kept bytes first (ascending), then the remapped ones (ascending)
    order = sorted(keep) + [b for b in range(256) if b not in keep]
    return table, [table[b] for b in order]


class ClipTokenizer:
    def __init__(self, bpe_path: str):
        if not bpe_path:
            raise ValueError("ClipTokenizer needs the path of a local bpe_simple_vocab_16e6.txt.gz (no download here)")
        self.byte_encoder, base = byte_symbols()
        opener = gzip.open if str(bpe_path).endswith(".gz") else open
        with opener(bpe_path, "rb") as f:
            lines = f.read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:_N_MERGES + 1]]
        vocab = base + [s + _END for s in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {s: i for i, s in enumerate(vocab)}
        if len(self.encoder) != len(vocab):                       # duplicate strings keep the LAST id, as dict(zip(...)) does
            self.encoder = dict(zip(vocab, range(len(vocab))))
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = _WORD

    # -- one word -> space-joined BPE symbols (same return convention as tokenizer/clip.py:27-66)
    def bpe(self, token):
        hit = self.cache.get(token)
        if hit is not None:
            return hit
        syms = list(token[:-1]) + [token[-1] + _END]
        if len(syms) == 1:
            return token + _END                                   # (:33-34: not cached, same as the reference)
        ranks = self.bpe_ranks
        while len(syms) > 1:
            best, best_rank = None, None
            for pair in zip(syms, syms[1:]):
                r = ranks.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            a, b = best
            merged, i, n = [], 0, len(syms)
            while i < n:
                if i + 1 < n and syms[i] == a and syms[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(syms[i])
                    i += 1
            syms = merged
        out = " ".join(syms)
        self.cache[token] = out
        return out

    def encode(self, text):
        """-> list of 77 ids: start, at most 75 BPE ids, end-of-text padding (tokenizer/clip.py:68-78)."""
        text = re.sub(r"\s+", " ", text.strip()).strip().lower()
        ids = []
        for word in self.pat.findall(text):
            word = "".join(self.byte_encoder[b] for b in word.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self.bpe(word).split(" "))
        ids = ids[:_CONTEXT - 2]
        return [_START_ID] + ids + [_END_ID] * (_CONTEXT - 1 - len(ids))
