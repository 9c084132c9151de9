"""Character tokenizer of the sampler front end (reference tokenizer.py:7-34): 71-char
alphabet -> ids 2..72, 0 = pad, 1 = end-of-sentence, unknown -> 2 ('_')."""
from __future__ import annotations

import string


class Tokenizer:
    def __init__(self):
        self.text = "_" + string.ascii_letters + string.digits + ".?!,'\"- "
        self.tokens = {ch: i + 2 for i, ch in enumerate(self.text)}
        self.chars = {i + 2: ch for i, ch in enumerate(self.text)}
        self.chars[0], self.chars[1] = " ", "<end>"
        self.vocab_size = len(self.text) + 2

    def encode(self, text: str) -> list[int]:
        return [self.tokens.get(ch, 2) for ch in text] + [1]

    def decode(self, tokens) -> str:
        return "".join(self.chars[int(t)] for t in tokens)


def stroke_length(n_tokens: int) -> int:
    """L heuristic of infer() (reference inference.py:77-78): 16 per token, next multiple of 8 (+8)."""
    t = n_tokens * 16
    return t - (t % 8) + 8
