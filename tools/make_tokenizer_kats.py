"""Generate tests/golden/tokenizer_kats.json by running the REFERENCE's own tokenizer
(/root/reference/utils/clip_tokenizer.py, stdlib + numpy only) in this container.
Run from the repo root:  python tools/make_tokenizer_kats.py
"""
import json
import os
import sys

sys.path.insert(0, "/root/reference")
from utils.clip_tokenizer import SimpleTokenizer  # noqa: E402  (reference code, imported not copied)

TEXTS = [
    "ferrari f40", "text here", "a person walking a dog", "", "   leading and   trailing   ", "A RED Car, parked!",
    "it's the dog's ball; they've gone & we'll see", "3 men 42 dogs 2024", "naïve café — señor", "日本語のテキスト",
    "hello&amp;world &lt;b&gt;", "e-mail: someone@example.com!!!", "don't can't I'm you'd", "emoji 🚗 car",
    "white van with ladder on roof near the gate at night", "supercalifragilisticexpialidocious", "x" * 40, "$9.99 (50% off)",
]

if __name__ == "__main__":
    tok = SimpleTokenizer()
    out = {"provenance": "reference utils/clip_tokenizer.py::SimpleTokenizer.encode run in-container",
           "vocab_size": tok.vocab_size, "sot": tok.sot_token_id, "eot": tok.eot_token_id,
           "cases": [{"text": t, "ids": [int(i) for i in tok.encode(t)]} for t in TEXTS]}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tokenizer_kats.json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=True, indent=1)
    print("wrote", path, len(TEXTS), "cases; ferrari f40 ->", out["cases"][0]["ids"])
