"""Meat-cut abbreviation expansion (reference src/abbreviation_translator.py:11-105).

Matching is case-sensitive on whole words, longest abbreviation first, exactly as
in the reference (:77-83).  Note the consequence the reference lives with: the
data pipeline lower-cases a description BEFORE expanding it
(src/data_processing.py:63) and every key here carries a capital, so on that path
nothing is ever expanded.  The behaviour is kept because embeddings, ids and the
stored documents depend on the exact strings.
"""
import re

_ABBREVIATIONS = (
    ("Bn-in", "Bone in"), ("Bnls", "Boneless"), ("Cntr Cut", "Center Cut"), ("Cov", "Cover"),
    ("Dkle", "Deckle"), ("Dfatd", "Defatted"), ("Dnd", "Denuded"), ("Dia", "Diamond"),
    ("Div", "Divided"), ("Ex", "Extra"), ("Fr", "Fresh"), ("Frz", "Frozen"), ("Grnd", "Ground"),
    ("Inter", "Intermediate"), ("IM", "Individual Muscle"), ("Nk-off", "Neck off"),
    ("NTE", "Not to Exceed"), ("Oven-Prep", "Oven-Prepared"), ("Part", "Partially"),
    ("Pld", "Peeled"), ("Prthse", "Porterhouse"), ("Portn", "Portion"), ("Reg", "Regular"),
    ("Rst-Rdy", "Roast-Ready"), ("Rst", "Roast"), ("Rnd", "Round"), ("Sh Cut", "Short Cut"),
    ("Shld", "Shoulder"), ("Sirln", "Sirloin"), ("Sknd", "Skinned"), ("Sp", "Special"),
    ("Sq-Cut", "Square Cut"), ("Stk", "Steak"), ("Tender", "Tenderloin"),
    ("Tri Tip", "Triangle Tip"), ("Trmd", "Trimmed"), ("Untrmd", "Untrimmed"),
)

# longest first; Python's sort is stable, so equal lengths keep the table order (reference :77)
_ORDERED = sorted(_ABBREVIATIONS, key=lambda kv: len(kv[0]), reverse=True)
_PATTERNS = [(re.compile(r"\b" + re.escape(a) + r"\b"), full) for a, full in _ORDERED]


def get_abbreviation_map():
    return dict(_ABBREVIATIONS)


def expand_abbreviations(text):
    """Sequential whole-word substitution, one abbreviation at a time (so an
    expansion can itself be rewritten by a later, shorter key: 'Tender' ->
    'Tenderloin' never re-matches, 'Rst-Rdy' is consumed before 'Rst')."""
    if not text or not isinstance(text, str):
        return text
    for pat, full in _PATTERNS:
        text = pat.sub(full, text)
    return text


def expand_abbreviations_in_dataframe(df, text_columns):
    out = df.copy()
    for col in text_columns:
        if col in out.columns:
            out[col] = out[col].apply(expand_abbreviations)
    return out
