"""Tokenizer for the whisper_amd host side: a self-contained byte-pair encoder (no tiktoken dependency)
plus the special-token bookkeeping the decoding loop needs.

Mirrors the interface of the reference's `whisper/tokenizer.py` (Tokenizer :131-327, get_encoding :330-363,
get_tokenizer :366-395): same attribute names, same token ids, because DecodingTask / transcribe /
find_alignment address tokens by those names.  The BPE itself is restated from tiktoken's published
algorithm (regex pre-split, then greedy lowest-rank pair merging over bytes).  Vocabulary ranks are
data, vendored gzip-compressed under whisper_amd/assets/ (provenance: openai/whisper assets, MIT).
"""
from __future__ import annotations

import base64
import gzip
import os
import string
from dataclasses import dataclass, field
from functools import cached_property, lru_cache
from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple

import regex

_LANG_TABLE = (
    "en:english;zh:chinese;de:german;es:spanish;ru:russian;ko:korean;fr:french;ja:japanese;pt:portuguese;"
    "tr:turkish;pl:polish;ca:catalan;nl:dutch;ar:arabic;sv:swedish;it:italian;id:indonesian;hi:hindi;"
    "fi:finnish;vi:vietnamese;he:hebrew;uk:ukrainian;el:greek;ms:malay;cs:czech;ro:romanian;da:danish;"
    "hu:hungarian;ta:tamil;no:norwegian;th:thai;ur:urdu;hr:croatian;bg:bulgarian;lt:lithuanian;la:latin;"
    "mi:maori;ml:malayalam;cy:welsh;sk:slovak;te:telugu;fa:persian;lv:latvian;bn:bengali;sr:serbian;"
    "az:azerbaijani;sl:slovenian;kn:kannada;et:estonian;mk:macedonian;br:breton;eu:basque;is:icelandic;"
    "hy:armenian;ne:nepali;mn:mongolian;bs:bosnian;kk:kazakh;sq:albanian;sw:swahili;gl:galician;mr:marathi;"
    "pa:punjabi;si:sinhala;km:khmer;sn:shona;yo:yoruba;so:somali;af:afrikaans;oc:occitan;ka:georgian;"
    "be:belarusian;tg:tajik;sd:sindhi;gu:gujarati;am:amharic;yi:yiddish;lo:lao;uz:uzbek;fo:faroese;"
    "ht:haitian creole;ps:pashto;tk:turkmen;nn:nynorsk;mt:maltese;sa:sanskrit;lb:luxembourgish;my:myanmar;"
    "bo:tibetan;tl:tagalog;mg:malagasy;as:assamese;tt:tatar;haw:hawaiian;ln:lingala;ha:hausa;ba:bashkir;"
    "jw:javanese;su:sundanese;yue:cantonese"
)
# language code -> name, in token-id order (the id of <|xx|> is sot + 1 + index)
LANGUAGES: Dict[str, str] = dict(item.split(":") for item in _LANG_TABLE.split(";"))
# name / alias -> code
TO_LANGUAGE_CODE: Dict[str, str] = {
    **{name: code for code, name in LANGUAGES.items()},
    "burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb",
    "pushto": "ps", "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si",
    "castilian": "es", "mandarin": "zh",
}

_SPLIT_PATTERN = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


class Encoding:
    """Minimal byte-pair encoder with the slice of tiktoken.Encoding's surface the path uses."""

    def __init__(self, name: str, ranks: Dict[bytes, int], special_tokens: Dict[str, int]):
        self.name = name
        self._ranks = ranks
        self._special = dict(special_tokens)
        self._bytes_of: Dict[int, bytes] = {r: b for b, r in ranks.items()}
        for s, i in self._special.items():
            self._bytes_of[i] = s.encode("utf-8")
        self._split = regex.compile(_SPLIT_PATTERN)
        self._special_re = regex.compile("|".join(regex.escape(s) for s in sorted(self._special, key=len, reverse=True)))
        self.n_vocab = len(ranks) + len(special_tokens)
        self.eot_token = self._special["<|endoftext|>"]

    @property
    def special_tokens_set(self) -> Set[str]:
        return set(self._special)

    def encode_single_token(self, text: str) -> int:
        if text in self._special:
            return self._special[text]
        return self._ranks[text.encode("utf-8")]

    def decode_single_token_bytes(self, token: int) -> bytes:
        return self._bytes_of[token]

    def _merge(self, piece: bytes) -> List[int]:
        if piece in self._ranks:
            return [self._ranks[piece]]
        parts = [piece[i: i + 1] for i in range(len(piece))]
        while len(parts) > 1:
            best_rank, best_i = None, -1
            for i in range(len(parts) - 1):
                r = self._ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best_rank is None or r < best_rank):
                    best_rank, best_i = r, i
            if best_rank is None:
                break
            parts[best_i: best_i + 2] = [parts[best_i] + parts[best_i + 1]]
        return [self._ranks[p] for p in parts]

    def _encode_ordinary(self, text: str) -> List[int]:
        out: List[int] = []
        for piece in self._split.findall(text):
            out.extend(self._merge(piece.encode("utf-8")))
        return out

    def encode(self, text: str, *, allowed_special=(), disallowed_special="all") -> List[int]:
        allowed = self.special_tokens_set if allowed_special == "all" else set(allowed_special)
        if disallowed_special == "all":
            disallowed = self.special_tokens_set - allowed
        else:
            disallowed = set(disallowed_special)
        if disallowed:
            m = self._special_re.search(text)
            while m is not None:
                if m.group() in disallowed:
                    raise ValueError(f"Encountered text corresponding to disallowed special token {m.group()!r}")
                m = self._special_re.search(text, m.end())
        if not allowed:
            return self._encode_ordinary(text)
        out: List[int] = []
        pos = 0
        for m in self._special_re.finditer(text):
            if m.group() not in allowed:
                continue
            out.extend(self._encode_ordinary(text[pos: m.start()]))
            out.append(self._special[m.group()])
            pos = m.end()
        out.extend(self._encode_ordinary(text[pos:]))
        return out

    def decode_bytes(self, tokens: Iterable[int]) -> bytes:
        return b"".join(self._bytes_of[int(t)] for t in tokens)

    def decode(self, tokens: Iterable[int], errors: str = "replace") -> str:
        return self.decode_bytes(tokens).decode("utf-8", errors=errors)


def _vocab_file(name: str) -> str:
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
    candidates = [os.path.join(here, f"{name}.bpe.gz"), os.path.join(here, f"{name}.tiktoken")]
    try:   # an installed reference package also carries the rank files
        import importlib.util
        spec = importlib.util.find_spec("whisper")
        if spec is not None and spec.origin:
            candidates.append(os.path.join(os.path.dirname(spec.origin), "assets", f"{name}.tiktoken"))
    except Exception:
        pass
    for c in candidates:
        if os.path.isfile(c):
            return c
    raise FileNotFoundError(f"BPE vocabulary '{name}' not found (looked in {candidates})")


def _load_ranks(path: str) -> Dict[bytes, int]:
    opener = gzip.open if path.endswith(".gz") else open
    ranks: Dict[bytes, int] = {}
    with opener(path, "rb") as f:
        for line in f.read().splitlines():
            if line:
                tok, rank = line.split()
                ranks[base64.b64decode(tok)] = int(rank)
    return ranks


@lru_cache(maxsize=None)
def get_encoding(name: str = "gpt2", num_languages: int = 99) -> Encoding:
    """Ranks + the special tokens appended after them, in the id order of tokenizer.py:338-354."""
    ranks = _load_ranks(_vocab_file(name))
    specials = (["<|endoftext|>", "<|startoftranscript|>"]
                + [f"<|{lang}|>" for lang in list(LANGUAGES)[:num_languages]]
                + ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>",
                   "<|notimestamps|>"]
                + [f"<|{i * 0.02:.2f}|>" for i in range(1501)])
    n = len(ranks)
    return Encoding(name=os.path.basename(name), ranks=ranks,
                    special_tokens={s: n + i for i, s in enumerate(specials)})


@dataclass
class Tokenizer:
    """Special-token accessors around an Encoding (reference: tokenizer.py:131-327)."""

    encoding: Encoding
    num_languages: int
    language: Optional[str] = None
    task: Optional[str] = None
    sot_sequence: Tuple[int, ...] = ()
    special_tokens: Dict[str, int] = field(default_factory=dict)

    def __post_init__(self):
        for s in self.encoding.special_tokens_set:
            self.special_tokens[s] = self.encoding.encode_single_token(s)
        seq = [self.sot]
        if self.language is not None:
            seq.append(self.sot + 1 + tuple(LANGUAGES)[: self.num_languages].index(self.language))
        if self.task is not None:
            seq.append(self.transcribe if self.task == "transcribe" else self.translate)
        self.sot_sequence = tuple(seq)

    def encode(self, text, **kwargs):
        return self.encoding.encode(text, **kwargs)

    def decode(self, token_ids: Sequence[int], **kwargs) -> str:
        """text only: timestamp tokens (ids >= timestamp_begin) are dropped"""
        tb = self.timestamp_begin
        return self.encoding.decode([t for t in token_ids if t < tb], **kwargs)

    def decode_with_timestamps(self, token_ids: Sequence[int], **kwargs) -> str:
        return self.encoding.decode(token_ids, **kwargs)

    @cached_property
    def eot(self) -> int:
        return self.encoding.eot_token

    @cached_property
    def transcribe(self) -> int:
        return self.special_tokens["<|transcribe|>"]

    @cached_property
    def translate(self) -> int:
        return self.special_tokens["<|translate|>"]

    @cached_property
    def sot(self) -> int:
        return self.special_tokens["<|startoftranscript|>"]

    @cached_property
    def sot_lm(self) -> int:
        return self.special_tokens["<|startoflm|>"]

    @cached_property
    def sot_prev(self) -> int:
        return self.special_tokens["<|startofprev|>"]

    @cached_property
    def no_speech(self) -> int:
        return self.special_tokens["<|nospeech|>"]

    @cached_property
    def no_timestamps(self) -> int:
        return self.special_tokens["<|notimestamps|>"]

    @cached_property
    def timestamp_begin(self) -> int:
        return self.special_tokens["<|0.00|>"]

    @cached_property
    def language_token(self) -> int:
        if self.language is None:
            raise ValueError("This tokenizer does not have language token configured")
        return self.to_language_token(self.language)

    def to_language_token(self, language: str) -> int:
        token = self.special_tokens.get(f"<|{language}|>")
        if token:
            return token
        raise KeyError(f"Language {language} not found in tokenizer.")

    @cached_property
    def all_language_tokens(self) -> Tuple[int, ...]:
        ids = [i for s, i in self.special_tokens.items() if s.strip("<|>") in LANGUAGES]
        return tuple(sorted(ids))[: self.num_languages]

    @cached_property
    def all_language_codes(self) -> Tuple[str, ...]:
        return tuple(self.decode([t]).strip("<|>") for t in self.all_language_tokens)

    @cached_property
    def sot_sequence_including_notimestamps(self) -> Tuple[int, ...]:
        return tuple(self.sot_sequence) + (self.no_timestamps,)

    @cached_property
    def non_speech_tokens(self) -> Tuple[int, ...]:
        """ids suppressed by "-1" in DecodingOptions.suppress_tokens: speaker tags, music notes, brackets…
        Same symbol inventory and single-token rule as the reference (tokenizer.py:241-275)."""
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        music = set("♩♪♫♬♭♮♯")   # U+2640..U+267F: first byte-pair token is safe to suppress
        keep = {self.encoding.encode(" -")[0], self.encoding.encode(" '")[0]}
        for sym in symbols + list(music):
            for toks in (self.encoding.encode(sym), self.encoding.encode(" " + sym)):
                if len(toks) == 1 or sym in music:
                    keep.add(toks[0])
        return tuple(sorted(keep))

    # ---- word splitting (used by word timestamps, timing.py:218) ----------------------------------
    def split_to_word_tokens(self, tokens: List[int]):
        if self.language in {"zh", "ja", "th", "lo", "my", "yue"}:
            return self.split_tokens_on_unicode(tokens)     # no spaces: split wherever bytes decode cleanly
        return self.split_tokens_on_spaces(tokens)

    def split_tokens_on_unicode(self, tokens: List[int]):
        full = self.decode_with_timestamps(tokens)
        bad = "�"
        words, word_tokens, cur, offset = [], [], [], 0
        for tok in tokens:
            cur.append(tok)
            text = self.decode_with_timestamps(cur)
            if bad not in text or full[offset + text.index(bad)] == bad:
                words.append(text)
                word_tokens.append(cur)
                cur = []
                offset += len(text)
        return words, word_tokens

    def split_tokens_on_spaces(self, tokens: List[int]):
        subwords, subword_tokens = self.split_tokens_on_unicode(tokens)
        words, word_tokens = [], []
        for sub, toks in zip(subwords, subword_tokens):
            starts_word = (toks[0] >= self.eot or sub.startswith(" ") or sub.strip() in string.punctuation
                           or not words)
            if starts_word:
                words.append(sub)
                word_tokens.append(toks)
            else:
                words[-1] += sub
                word_tokens[-1].extend(toks)
        return words, word_tokens


@lru_cache(maxsize=None)
def get_tokenizer(multilingual: bool, *, num_languages: int = 99, language: Optional[str] = None,
                  task: Optional[str] = None) -> Tokenizer:
    if language is not None:
        language = language.lower()
        if language not in LANGUAGES:
            if language in TO_LANGUAGE_CODE:
                language = TO_LANGUAGE_CODE[language]
            else:
                raise ValueError(f"Unsupported language: {language}")
    if multilingual:
        name, language, task = "multilingual", language or "en", task or "transcribe"
    else:
        name, language, task = "gpt2", None, None
    return Tokenizer(encoding=get_encoding(name=name, num_languages=num_languages),
                     num_languages=num_languages, language=language, task=task)
