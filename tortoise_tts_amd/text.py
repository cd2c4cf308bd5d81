"""Text front-end: the reference's VoiceBpeTokenizer contract (tortoise/utils/tokenizer.py:172-185)
on top of the HF `tokenizers` BPE file the reference ships (tortoise/data/tokenizer.json, 255
entries).  The vocabulary file is data the user already has with the reference checkpoints; it is
located, not copied: explicit path -> $TORTOISE_TOKENIZER -> <models_dir>/tokenizer.json ->
an installed `tortoise` package's data directory.

CPU string work is outside the hot path; english_cleaners needs `inflect` and
`unidecode` exactly like the reference and is refused loudly when they are missing rather than
silently changing the token stream.
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
        if not use_basic_cleaners:
            try:
                import inflect  # noqa: F401
                import unidecode  # noqa: F401
            except ImportError as e:
                raise ImportError("english_cleaners needs `inflect` and `unidecode` (as in the reference); install them or "
                                  "construct TextToSpeech(tokenizer_basic=True)") from e

    def preprocess_text(self, txt):
        if self.use_basic:
            return basic_cleaners(txt)
        from .text_english import english_cleaners
        return english_cleaners(txt)

    def encode(self, txt):
        txt = self.preprocess_text(txt)
        txt = txt.replace(" ", "[SPACE]")
        return self.tokenizer.encode(txt).ids

    def decode(self, seq):
        if hasattr(seq, "cpu"):
            seq = seq.cpu().numpy()
        txt = self.tokenizer.decode(seq, skip_special_tokens=False).replace(" ", "")
        return txt.replace("[SPACE]", " ").replace("[STOP]", "").replace("[UNK]", "")
