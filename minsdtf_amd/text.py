"""Host-side text front-end of the pipeline: CLIP byte-pair tokenizer and prompt weighting.

These are the pieces between a prompt string and the token ids that the device-resident CLIP text
transformer (models.TextClipEmbedding / TextEncoder) consumes; they mirror the behaviour of the
reference's ``clip_tokenizer.SimpleTokenizer`` (clip_tokenizer.py:77-209) and
``long_prompt_weighting.get_weighted_text_embeddings`` (long_prompt_weighting.py:35-333) — same token
ids, same chunking into 77-token windows, same weighting and mean preservation — so that
``StableDiffusion.text_to_image("a (very:1.2) nice prompt")`` behaves like the reference's.  Nothing here
is on the hot path (it runs once per prompt on the host).

The tokenizer needs CLIP's merge list (``bpe_simple_vocab_16e6.txt.gz``), which the reference downloads;
here it must be given as a local path (``StableDiffusion(..., bpe_path=...)`` or ``$MSD_BPE_PATH``).
"""
from __future__ import annotations

import gzip
import html
import re as _re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

CLIP_VOCAB_SIZE = 49408
BOS, EOS = "<|startoftext|>", "<|endoftext|>"


def byte_alphabet() -> Dict[int, str]:
    """The reversible byte -> printable-unicode table of GPT-2 / CLIP BPE: printable latin-1 bytes map to
    themselves, the 68 others to code points 256, 257, ... in byte order (clip_tokenizer.py:25-49)."""
    printable = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    # the reference enumerates the printable bytes first, then the rest: dict ORDER defines the vocabulary order
    table = {b: chr(b) for b in printable}
    extra = 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class SimpleTokenizer:
    """CLIP's lower-cased byte-level BPE.  ``encode(text)`` -> [start] + ids + [end]."""

    def __init__(self, bpe_path: str):
        import regex

        self._regex = regex
        self.byte_encoder = byte_alphabet()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        with gzip.open(bpe_path) as f:
            lines = f.read().decode("utf-8").split("\n")
        merges = [tuple(line.split()) for line in lines[1: CLIP_VOCAB_SIZE - 256 - 2 + 1]]   # header line skipped
        symbols = list(self.byte_encoder.values())
        self.vocab: List[str] = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges] + [BOS, EOS]
        self.rank: Dict[Tuple[str, ...], int] = {m: i for i, m in enumerate(merges)}
        self.special_tokens: Dict[str, str] = {BOS: BOS, EOS: EOS}
        self.cache: Dict[str, str] = {BOS: BOS, EOS: EOS}
        self._reindex()

    # the reference exposes bpe_ranks under this name
    @property
    def bpe_ranks(self):
        return self.rank

    def _reindex(self) -> None:
        self.encoder = {tok: i for i, tok in enumerate(self.vocab)}     # (a repeated entry keeps its LAST index, like dict(zip()))
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        specials = "|".join(self._regex.escape(k) for k in self.special_tokens)
        self.pat = self._regex.compile(specials + r"""|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                       self._regex.IGNORECASE)

    @property
    def start_of_text(self) -> int:
        return self.encoder[BOS]

    @property
    def end_of_text(self) -> int:
        return self.encoder[EOS]

    def add_tokens(self, tokens) -> int:
        """New special tokens (textual inversion placeholders); returns how many were new (clip_tokenizer.py:130-144)."""
        if isinstance(tokens, str):
            tokens = [tokens]
        added = 0
        for tok in tokens:
            if tok in self.vocab:
                continue
            added += 1
            self.vocab.append(tok)
            self.special_tokens[tok] = tok
            self.cache[tok] = tok
        self._reindex()
        return added

    def bpe(self, token: str) -> str:
        """Merge the symbols of one pre-token, lowest-rank pair first, all its occurrences left to right."""
        hit = self.cache.get(token)
        if hit is not None:
            return hit
        word = list(token[:-1]) + [token[-1] + "</w>"]
        if len(word) < 2:
            return token + "</w>"          # (single symbol: not cached by the reference either)
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word, word[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    merged.append(best[0] + best[1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(text)).strip()
        text = self._regex.sub(r"\s+", " ", text).strip().lower()
        ids: List[int] = []
        for piece in self._regex.findall(self.pat, text):
            mapped = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[sym] for sym in self.bpe(mapped).split(" "))
        return [self.start_of_text] + ids + [self.end_of_text]

    def decode(self, tokens: Sequence[int]) -> str:
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


# ------------------------------------------------------------------------------------------------
# prompt weighting
# ------------------------------------------------------------------------------------------------
_ATTN = _re.compile(r"""
    (?P<esc>\\[()\[\]\\])      |   # escaped bracket or backslash -> literal
    (?P<bs>\\)                 |   # lone backslash (dropped, leaves an empty piece)
    (?P<open>[(\[])            |
    :(?P<w>[+-]?[.\d]+)\)      |   # explicit weight closing a round bracket
    (?P<close>[)\]])           |
    (?P<text>[^\\()\[\]:]+)    |
    (?P<colon>:)
""", _re.X)
ROUND_MULT, SQUARE_MULT = 1.1, 1 / 1.1


def parse_prompt_attention(text: str) -> List[list]:
    """"a (b) [c] (d:1.3)" -> [[piece, weight], ...]: round brackets x1.1, square brackets /1.1, (x:w) xw,
    backslash escapes, unbalanced openers closed at the end, equal-weight neighbours merged
    (long_prompt_weighting.py:35-110)."""
    out: List[list] = []
    opened = {"(": [], "[": []}

    def scale_from(start: int, factor: float) -> None:
        for item in out[start:]:
            item[1] *= factor

    for m in _ATTN.finditer(text):
        kind = m.lastgroup
        if kind == "esc" or kind == "bs":
            out.append([m.group(0)[1:], 1.0])
        elif kind == "open":
            opened[m.group(0)].append(len(out))
        elif kind == "w" and opened["("]:
            scale_from(opened["("].pop(), float(m.group("w")))
        elif kind == "close" and m.group(0) == ")" and opened["("]:
            scale_from(opened["("].pop(), ROUND_MULT)
        elif kind == "close" and m.group(0) == "]" and opened["["]:
            scale_from(opened["["].pop(), SQUARE_MULT)
        else:
            out.append([m.group(0), 1.0])
    for start in opened["("]:
        scale_from(start, ROUND_MULT)
    for start in opened["["]:
        scale_from(start, SQUARE_MULT)
    if not out:
        out = [["", 1.0]]
    merged = [out[0]]
    for piece, w in out[1:]:
        if w == merged[-1][1]:
            merged[-1][0] += piece
        else:
            merged.append([piece, w])
    return merged


def tokens_with_weights(tokenizer, prompts: Sequence[str], limit: int, embedding_tokens_count: int = 0,
                        embedding_tokens_weight: float = 1.0):
    """Per prompt: token ids (no start / end / padding) and one weight per token, cut at `limit`
    (long_prompt_weighting.py:113-154)."""
    all_tokens, all_weights, truncated = [], [], False
    for text in prompts:
        toks: List[int] = []
        wts: List[float] = []
        if embedding_tokens_count > 0:   # placeholders that the textual-inversion vectors overwrite later
            toks += tokenizer.encode("*")[1:-1] * embedding_tokens_count
            wts += [embedding_tokens_weight] * embedding_tokens_count
        for piece, weight in parse_prompt_attention(text):
            ids = list(tokenizer.encode(piece.strip())[1:-1])
            toks += ids
            wts += [weight] * len(ids)
            if len(toks) > limit:
                break
        if len(toks) > limit:
            truncated = True
            toks, wts = toks[:limit], wts[:limit]
        all_tokens.append(toks)
        all_weights.append(wts)
    if truncated:
        print("Prompt was truncated. Try to shorten the prompt or increase max_embeddings_multiples")
    return all_tokens, all_weights


def pad_tokens_and_weights(tokens, weights, max_length, bos, eos, pad, no_boseos_middle=True, chunk_length=77):
    """[bos] tokens [pad]* [eos] of length max_length; weights 1.0 on the added positions — for
    no_boseos_middle=False also on the start / end slot of every 77-token window (long_prompt_weighting.py:157-181)."""
    windows = (max_length - 2) // (chunk_length - 2)
    body = chunk_length - 2
    out_t, out_w = [], []
    for toks, wts in zip(tokens, weights):
        out_t.append([bos] + toks + [pad] * (max_length - 2 - len(toks)) + [eos])
        if no_boseos_middle:
            out_w.append([1.0] + wts + [1.0] * (max_length - 1 - len(wts)))
        elif not wts:
            out_w.append([1.0] * (windows * chunk_length))
        else:
            w: List[float] = []
            for j in range(windows):
                w += [1.0] + wts[j * body: min(len(wts), (j + 1) * body)] + [1.0]
            out_w.append(w + [1.0] * (windows * chunk_length - len(w)))
    return out_t, out_w


def _inject(clip_embedding, embedding, count):
    """Textual inversion: the `count` placeholder positions after the start token take the learned vectors."""
    return np.concatenate([clip_embedding[:, 0:1, :], np.tile(embedding, (clip_embedding.shape[0], 1, 1)).astype(clip_embedding.dtype),
                           clip_embedding[:, count + 1:, :]], axis=1)


def unweighted_text_embeddings(text_clip_embedding, text_encoder, token_ids: np.ndarray, chunk_length: int,
                               no_boseos_middle: bool = True, embedding_tokens_count: int = 0, embedding=None):
    """Encode (B, 75k+2) token ids window by window: window i = ids[75i : 75i+77] with the global start / end
    token written into its first / last slot (long_prompt_weighting.py:184-239)."""
    inject = embedding_tokens_count > 0 and embedding is not None
    windows = (token_ids.shape[1] - 2) // (chunk_length - 2)

    def encode(ids, with_embedding):
        pos = np.asarray([list(range(ids.shape[1]))], dtype=np.int32)
        emb = text_clip_embedding.predict_on_batch([ids, pos])
        if with_embedding:
            emb = _inject(emb, embedding, embedding_tokens_count)
        return text_encoder.predict_on_batch(emb)

    if windows <= 1:
        return encode(token_ids, inject)
    body = chunk_length - 2
    parts = []
    for i in range(windows):
        ids = token_ids[:, i * body: (i + 1) * body + 2].copy()
        ids[:, 0], ids[:, -1] = token_ids[0, 0], token_ids[0, -1]
        part = encode(ids, inject and i == 0)
        if no_boseos_middle:
            part = part[:, :-1] if i == 0 else (part[:, 1:] if i == windows - 1 else part[:, 1:-1])
        parts.append(part)
    return np.concatenate(parts, axis=1)


def get_weighted_text_embeddings(tokenizer, text_clip_embedding, text_encoder, prompt, max_embeddings_multiples: Optional[int] = 4,
                                 no_boseos_middle: Optional[bool] = False, skip_parsing: Optional[bool] = False,
                                 skip_weighting: Optional[bool] = False, model_max_length=77, pad_token_id=49407,
                                 embedding_tokens_count=0, embedding_tokens_weight=1.0, embedding=None):
    """Prompt(s) -> (B, 77k, 768) context with per-token emphasis, mean-preserving
    (long_prompt_weighting.py:242-333; same signature and defaults)."""
    if embedding_tokens_count > 0 and embedding is None:
        embedding_tokens_count = 0
    body = model_max_length - 2
    if isinstance(prompt, str):
        prompt = [prompt]
    if not skip_parsing:
        tokens, weights = tokens_with_weights(tokenizer, prompt, body * max_embeddings_multiples, embedding_tokens_count,
                                              embedding_tokens_weight)
    else:
        tokens = [tok[1:-1] for tok in tokenizer.encode(prompt)[:body * max_embeddings_multiples + 2]]
        weights = [[1.0] * len(tok) for tok in tokens]
    longest = max(len(t) for t in tokens)
    multiples = max(1, min(max_embeddings_multiples, (longest - 1) // body + 1))
    max_length = body * multiples + 2
    tokens, weights = pad_tokens_and_weights(tokens, weights, max_length, tokenizer.start_of_text, tokenizer.end_of_text,
                                             pad_token_id, no_boseos_middle=no_boseos_middle, chunk_length=model_max_length)
    ids = np.array(tokens, dtype=np.int32)
    emb = unweighted_text_embeddings(text_clip_embedding, text_encoder, ids, model_max_length, no_boseos_middle=no_boseos_middle,
                                     embedding_tokens_count=embedding_tokens_count, embedding=embedding)
    w = np.array(weights, dtype=emb.dtype)
    if (not skip_parsing) and (not skip_weighting):
        before = emb.mean(axis=(-2, -1))
        emb *= w[:, :, None]
        emb *= (before / emb.mean(axis=(-2, -1)))[:, None, None]
    return emb
