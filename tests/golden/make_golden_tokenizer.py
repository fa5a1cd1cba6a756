"""Golden token ids from the REFERENCE's own tokenizer (src/open_clip/tokenizer.py SimpleTokenizer).

Run in the BUILD container only:  python tests/golden/make_golden_tokenizer.py
The reference module needs ftfy (not installed here); it is stubbed with the identity -- every text below is plain
ASCII / already-clean unicode, for which ftfy.fix_text is the identity anyway.  Output: tokenizer_golden.json
(texts, their [n,77] token rows, and two rows at context length 16 to pin the truncation rule)."""
import importlib.util
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/src/open_clip/tokenizer.py"
OUT = os.path.dirname(os.path.abspath(__file__))

TEXTS = [
    "ACTB GAPDH MALAT1 TMSB4X RPLP0 EEF1A1 FTL B2M",
    "mt-co1 MT-ND4 rps27a; HLA-DRA (CD74) krt8/krt18",
    "a photo of a tissue tile, stained with H&amp;E &lt;20x&gt;",
    "  multiple   spaces\tand\nnewlines  ",
    "it's the patient's 3rd biopsy: ER+ PR- HER2 2+",
    "IGKC IGHG1 IGHA1 JCHAIN MZB1 XBP1 SSR4 DERL3 FKBP11 PRDX4 SEC11C HSP90B1 " * 6,
    "",
    "naïve café ß-catenin 細胞 12345",
    "<start_of_text> literal specials <end_of_text>",
    "COL1A1 COL1A2 COL3A1 SPARC FN1 DCN LUM POSTN",
]


def main():
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda t: t
    sys.modules["ftfy"] = ftfy
    spec = importlib.util.spec_from_file_location("ref_tokenizer", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tok = mod.SimpleTokenizer()
    full = tok(TEXTS).tolist()
    short = tok(TEXTS[:2] + TEXTS[5:6], context_length=16).tolist()
    json.dump({"texts": TEXTS, "ids": full, "short_texts": TEXTS[:2] + TEXTS[5:6], "short_ids": short,
               "vocab_size": tok.vocab_size, "sot": tok.sot_token_id, "eot": tok.eot_token_id},
              open(os.path.join(OUT, "tokenizer_golden.json"), "w"))
    print("wrote tokenizer_golden.json", tok.vocab_size, [len([t for t in r if t]) for r in full])


if __name__ == "__main__":
    main()
