"""Import shim so the *reference* package (/root/reference/whisper) can be imported in this container,
where the real `tiktoken` (a Rust extension) is not installed.  Test infrastructure only: it adapts the
constructor signature the reference uses (whisper/tokenizer.py:357-363) onto whisper_amd's own BPE."""
from whisper_amd.tokenizer import Encoding as _Enc


class Encoding(_Enc):
    def __init__(self, name, *, explicit_n_vocab=None, pat_str=None, mergeable_ranks=None, special_tokens=None):
        super().__init__(name=name, ranks=mergeable_ranks, special_tokens=special_tokens)
        if explicit_n_vocab is not None:
            assert explicit_n_vocab == self.n_vocab
