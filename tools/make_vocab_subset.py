"""One-off tool (needs the reference checkout): cut the merges the tokenizer KAT strings can ever look up out of clearcam's
``utils/bpe_simple_vocab_16e6.txt.gz`` -> tests/golden/bpe_merges_subset.json ({"left right": published rank}).

The GPU box has no clearcam checkout, so the string surface ``OpenCLIP._encode_text(str)`` could not be tested there.  With
this subset (a few hundred of the 48 894 merges, ranks as published) ``SimpleTokenizer(sparse_merges=...)`` tokenises exactly
the strings of tests/golden/tokenizer_kats.json (ids produced by the REFERENCE's tokenizer) and nothing else is claimed.
A merge is kept if any step of the greedy BPE loop over those strings finds it in the table (chosen or not): the loop's
decisions then cannot differ from the full table's."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clearcam_amd.clip_tokenizer import SimpleTokenizer  # noqa: E402

REF_VOCAB = "/root/reference/utils/bpe_simple_vocab_16e6.txt.gz"


class Recording(dict):
    def __init__(self, base):
        super().__init__(base)
        self.hits = {}

    def get(self, key, default=None):
        r = super().get(key, default)
        if r is not None:
            self.hits[key] = r
        return r


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kats = json.load(open(os.path.join(root, "tests", "golden", "tokenizer_kats.json")))
    t = SimpleTokenizer(REF_VOCAB)
    t.rank = Recording(t.rank)
    for case in kats["cases"]:
        assert t.encode(case["text"]) == case["ids"], case["text"]
    out = {f"{a} {b}": r for (a, b), r in sorted(t.rank.hits.items(), key=lambda kv: kv[1])}
    json.dump({"provenance": "merges of clearcam's utils/bpe_simple_vocab_16e6.txt.gz (OpenAI CLIP vocabulary) that the strings of "
                             "tokenizer_kats.json look up; ranks as published; tools/make_vocab_subset.py", "merges": out},
              open(os.path.join(root, "tests", "golden", "bpe_merges_subset.json"), "w"), ensure_ascii=True, indent=0)
    print(len(out), "merges kept")
