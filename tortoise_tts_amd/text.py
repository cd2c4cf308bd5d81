"""Text front-end: the reference's VoiceBpeTokenizer contract (tortoise/utils/tokenizer.py:172-185)
on top of the HF `tokenizers` BPE file the reference ships (tortoise/data/tokenizer.json, 255
entries).  The vocabulary file is data the user already has with the reference checkpoints; it is
located, not copied: explicit path -> $TORTOISE_TOKENIZER -> <models_dir>/tokenizer.json ->
an installed `tortoise` package's data directory.

CPU string work is outside the hot path ("reuse as-is"): `english_cleaners` is NOT
re-implemented here.  With tokenizer_basic=False the reference's own module (tortoise/utils/tokenizer.py:122-150,
which needs `inflect` and `unidecode`) is imported from an installed `tortoise` package; when it is not importable the
constructor refuses loudly instead of silently changing the token stream.
"""
import os
import re

_whitespace_re = re.compile(r"\s+")


def basic_cleaners(text):
    """tokenizer.py:127-131: lowercase + collapse whitespace."""
    return re.sub(_whitespace_re, " ", text.lower())


def find_vocab_file(explicit=None, models_dir=None):
    cands = [explicit, os.environ.get("TORTOISE_TOKENIZER")]
    if models_dir:
        cands.append(os.path.join(models_dir, "tokenizer.json"))
    try:
        import importlib.util
        spec = importlib.util.find_spec("tortoise")
        if spec and spec.submodule_search_locations:
            cands.append(os.path.join(list(spec.submodule_search_locations)[0], "data", "tokenizer.json"))
    except Exception:
        pass
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


class VoiceBpeTokenizer:
    def __init__(self, vocab_file=None, use_basic_cleaners=False, models_dir=None):
        path = find_vocab_file(vocab_file, models_dir)
        if path is None:
            raise FileNotFoundError("tokenizer.json not found: pass tokenizer_vocab_file=, set TORTOISE_TOKENIZER, or put it in "
                                    "models_dir (it ships with the reference at tortoise/data/tokenizer.json)")
        from tokenizers import Tokenizer
        self.tokenizer = Tokenizer.from_file(path)
        self.use_basic = use_basic_cleaners
        self._english_cleaners = None
        if not use_basic_cleaners:
            try:
                from tortoise.utils.tokenizer import english_cleaners  # the reference's own cleaners, reused as-is
            except ImportError as e:
                raise ImportError("english_cleaners come from the reference package (tortoise.utils.tokenizer, which needs `inflect` "
                                  "and `unidecode`); install it or construct TextToSpeech(tokenizer_basic=True)") from e
            self._english_cleaners = english_cleaners

    def preprocess_text(self, txt):
        if self.use_basic:
            return basic_cleaners(txt)
        return self._english_cleaners(txt)

    def encode(self, txt):
        txt = self.preprocess_text(txt)
        txt = txt.replace(" ", "[SPACE]")
        return self.tokenizer.encode(txt).ids

    def decode(self, seq):
        if hasattr(seq, "cpu"):
            seq = seq.cpu().numpy()
        txt = self.tokenizer.decode(seq, skip_special_tokens=False).replace(" ", "")
        return txt.replace("[SPACE]", " ").replace("[STOP]", "").replace("[UNK]", "")


# ------------------------------------------------------------------------------------------------
# Long-form chunking (the caller side of the hot path for read.py-style use, reference behaviour:
# tortoise/utils/text.py:4-72, pinned by the reference's own expectations at text.py:82-130).
# Own implementation as a cursor over the normalised string; `_Cursor` reproduces the reference's
# observable quirks, which its expectations depend on:
#   * a look-ahead that reaches the LAST character (or beyond) yields "" and "" matches every
#     character class (Python's `"" in "abc"` is True), so the final character always ends a sentence;
#   * stepping BACK onto a double quote toggles the in-quote state just like stepping forward onto it.
_SENTENCE_END = "!?\n"
_BREAKABLE = "!?.\n "


class _Cursor:
    def __init__(self, text):
        self.text = text
        self.pos = -1          # index of the last consumed character
        self.start = 0         # first index of the chunk being built
        self.in_quote = False

    def __len__(self):         # length of the chunk being built
        return self.pos - self.start + 1

    def _land(self):
        if self.text[self.pos] == '"':
            self.in_quote = not self.in_quote
        return self.text[self.pos]

    def forward(self, n=1):
        ch = ""
        for _ in range(n):
            self.pos += 1
            ch = self._land()
        return ch

    def back(self, n=1):
        ch = ""
        for _ in range(n):
            self.pos -= 1
            ch = self._land()
        return ch

    def ahead(self, k):
        p = self.pos + k
        return self.text[p] if 0 <= p < len(self.text) - 1 else ""

    def take(self):
        chunk = self.text[self.start:self.pos + 1]
        self.start = self.pos + 1
        return chunk


def _among(ch, chars):
    return ch == "" or ch in chars


def split_and_recombine_text(text, desired_length=200, max_length=300):
    """Split `text` into chunks of about `desired_length` characters (never more than `max_length`), preferring
    sentence boundaries and keeping quoted passages together."""
    text = re.sub(r"\n\n+", "\n", text)
    text = re.sub(r"\s+", " ", text)
    text = re.sub(r"[“”]", '"', text)
    cur = _Cursor(text)
    last = len(text) - 1
    chunks, boundaries = [], []   # boundaries: positions inside the current chunk where a sentence ended

    def emit():
        chunks.append(cur.take())
        boundaries.clear()

    while cur.pos < last:
        ch = cur.forward()
        if len(cur) >= max_length:
            # forced split: back to the last sentence end if the chunk is already half full, else to a word break
            if boundaries and len(cur) > desired_length / 2:
                cur.back(cur.pos - boundaries[-1])
            else:
                while not (ch in _BREAKABLE) and cur.pos > 0 and len(cur) > desired_length:
                    ch = cur.back()
            emit()
        elif not cur.in_quote and (ch in _SENTENCE_END or (ch == "." and _among(cur.ahead(1), "\n "))):
            # swallow runs of terminal punctuation ("?!?!", "....") while there is room
            while cur.pos < last and len(cur) < max_length and _among(cur.ahead(1), "!?."):
                ch = cur.forward()
            boundaries.append(cur.pos)
            if len(cur) >= desired_length:
                emit()
        elif cur.in_quote and cur.ahead(1) == '"' and _among(cur.ahead(2), "\n "):
            # a closing quote followed by whitespace also ends a sentence
            cur.forward(2)
            boundaries.append(cur.pos)
    chunks.append(cur.take())
    chunks = [c.strip() for c in chunks]
    return [c for c in chunks if c and not re.match(r"^[\s\.,;:!?]*$", c)]
