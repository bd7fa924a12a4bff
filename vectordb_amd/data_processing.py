"""Transaction-report pipeline feeding the index build (reference
src/data_processing.py:6-88): load the workbook, keep (code, description), clean,
expand, de-duplicate by cleaned description keeping the first code.

`read_excel` uses pandas' own reader when an xlsx engine is importable and
otherwise a small stdlib (zipfile + ElementTree) reader that returns what pandas
would for the plain tabular sheets this project holds: first row = header,
numeric cells int unless written with a fraction/exponent, shared/inline strings,
booleans, empty cells NaN.  Date styles are not interpreted (none of the
reference's sheets the pipeline reads carry date columns it uses).
"""
import re
import zipfile
import xml.etree.ElementTree as ET

import numpy as np
import pandas as pd

from . import config
from .abbreviation_translator import expand_abbreviations

_NS = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main",
       "r": "http://schemas.openxmlformats.org/officeDocument/2006/relationships",
       "p": "http://schemas.openxmlformats.org/package/2006/relationships"}
_COL_RE = re.compile(r"([A-Z]+)")
# pandas.read_excel's default NA strings (keep_default_na=True): such a TEXT cell becomes NaN.  The reference's
# transaction report holds two descriptions that are the literal text "None"; pandas drops them as missing, which
# is why its store has 14 777 products, not 14 778.
_NA_STRINGS = frozenset(["", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>",
                         "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"])


def _col_index(ref: str) -> int:
    n = 0
    for ch in _COL_RE.match(ref).group(1):
        n = n * 26 + (ord(ch) - 64)
    return n - 1


def _text_of(node) -> str:
    # <si>/<is> hold either one <t> or rich-text runs <r><t>; phonetic runs <rPh> are skipped
    parts = []
    for child in node:
        tag = child.tag.rsplit("}", 1)[-1]
        if tag == "t":
            parts.append(child.text or "")
        elif tag == "r":
            for t in child.findall("m:t", _NS):
                parts.append(t.text or "")
    return "".join(parts)


def _number(text: str):
    if "." in text or "e" in text or "E" in text:
        return float(text)
    return int(text)


def _read_xlsx_stdlib(path, sheet_name=0) -> pd.DataFrame:
    with zipfile.ZipFile(path) as z:
        wb = ET.fromstring(z.read("xl/workbook.xml"))
        rels = ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))
        targets = {r.get("Id"): r.get("Target") for r in rels.findall("p:Relationship", _NS)}
        sheets = [(s.get("name"), targets[s.get("{%s}id" % _NS["r"])]) for s in wb.findall("m:sheets/m:sheet", _NS)]
        if isinstance(sheet_name, int):
            if not 0 <= sheet_name < len(sheets):
                raise ValueError(f"Worksheet index {sheet_name} is invalid, {len(sheets)} worksheets found")
            target = sheets[sheet_name][1]
        else:
            match = [t for n, t in sheets if n == sheet_name]
            if not match:
                raise ValueError(f"Worksheet named '{sheet_name}' not found")
            target = match[0]
        target = target.lstrip("/")
        if not target.startswith("xl/"):
            target = "xl/" + target
        shared = []
        if "xl/sharedStrings.xml" in z.namelist():
            shared = [_text_of(si) for si in ET.fromstring(z.read("xl/sharedStrings.xml")).findall("m:si", _NS)]
        rows = []
        with z.open(target) as f:
            next_row = 0
            for _, row in ET.iterparse(f):
                if row.tag.rsplit("}", 1)[-1] != "row":
                    continue
                r_attr = row.get("r")
                r_idx = int(r_attr) - 1 if r_attr else next_row
                next_row = r_idx + 1
                cells = {}
                next_col = 0
                for c in row.findall("m:c", _NS):
                    ref = c.get("r")
                    ci = _col_index(ref) if ref else next_col
                    next_col = ci + 1
                    t = c.get("t", "n")
                    v = c.find("m:v", _NS)
                    val = None
                    if t == "inlineStr":
                        is_node = c.find("m:is", _NS)
                        val = _text_of(is_node) if is_node is not None else None
                    elif v is not None and v.text is not None:
                        if t == "s":
                            val = shared[int(v.text)]
                        elif t == "str":
                            val = v.text
                        elif t == "b":
                            val = v.text == "1"
                        elif t == "e":
                            val = None
                        else:
                            val = _number(v.text)
                    if val is not None and not (isinstance(val, str) and val in _NA_STRINGS):
                        cells[ci] = val
                rows.append((r_idx, cells))
                row.clear()
    rows = [(i, c) for i, c in rows if c]
    if not rows:
        return pd.DataFrame()
    width = 1 + max(max(c) for _, c in rows)
    first = rows[0][0]
    header_cells = rows[0][1]
    header, seen = [], {}
    for j in range(width):
        name = header_cells.get(j, f"Unnamed: {j}")
        if name in seen:                       # pandas' mangle_dupe_cols
            seen[name] += 1
            name = f"{name}.{seen[name]}"
        else:
            seen[name] = 0
        header.append(name)
    last = rows[-1][0]
    by_row = dict(rows[1:])
    cols = {j: [] for j in range(width)}
    for i in range(first + 1, last + 1):       # blank rows inside the used range are kept, as pandas keeps them
        cells = by_row.get(i, {})
        for j in range(width):
            cols[j].append(cells.get(j))
    data = {}
    for j, name in enumerate(header):
        vals = cols[j]
        kinds = {type(v) for v in vals if v is not None}
        if kinds and kinds <= {int, float}:
            if kinds == {int} and all(v is not None for v in vals):
                data[name] = np.asarray(vals, dtype=np.int64)
            else:
                data[name] = np.asarray([np.nan if v is None else v for v in vals], dtype=np.float64)
        elif kinds == {bool} and all(v is not None for v in vals):
            data[name] = np.asarray(vals, dtype=bool)
        else:
            arr = np.empty(len(vals), dtype=object)
            for i, v in enumerate(vals):
                arr[i] = np.nan if v is None else v
            data[name] = arr
    return pd.DataFrame(data, columns=header)


def read_excel(path, sheet_name=0) -> pd.DataFrame:
    """pandas.read_excel when it has an engine; the stdlib reader otherwise."""
    try:
        import openpyxl  # noqa: F401
    except ImportError:
        return _read_xlsx_stdlib(path, sheet_name)
    return pd.read_excel(path, sheet_name=sheet_name)


def clean_text(text):
    """Lower-case, strip, collapse inner whitespace; non-strings pass through (reference :6-12)."""
    if isinstance(text, str):
        text = re.sub(r"\s+", " ", text.lower().strip())
    return text


def load_transaction_data(file_path=None, sheet_name=None):
    """DataFrame, or None (with a message) when the file or sheet cannot be read (reference :14-27)."""
    file_path = config.TRANSACTION_REPORT_FILE if file_path is None else file_path
    sheet_name = config.TRANSACTION_SHEET_NAME if sheet_name is None else sheet_name
    print(f"Loading transaction data from: {file_path}, Sheet: {sheet_name}")
    try:
        df = read_excel(file_path, sheet_name=sheet_name)
    except FileNotFoundError:
        print(f"Error: Transaction file not found at {file_path}")
        return None
    except Exception as exc:
        print(f"Error loading transaction data: {exc}")
        return None
    print(f"Successfully loaded {len(df)} rows.")
    return df


def process_transaction_data(df_raw):
    """-> DataFrame[product_description, product_code], one row per distinct cleaned
    description in sorted order, carrying the first code seen for it (reference :29-88)."""
    if df_raw is None:
        print("No raw data to process.")
        return None
    code_col, desc_col = config.TRANSACTION_PRODUCT_CODE_COL, config.TRANSACTION_DESC_COL
    if code_col not in df_raw.columns or desc_col not in df_raw.columns:
        print(f"Error: Required columns ('{code_col}', '{desc_col}') not found in the dataframe.")
        print(f"Available columns: {df_raw.columns.tolist()}")
        return None
    df = df_raw[[code_col, desc_col]].dropna(subset=[code_col, desc_col]).copy()
    if len(df) < len(df_raw):
        print(f"Dropped {len(df_raw) - len(df)} rows with missing ProductCode or ProductDescription.")
    df[code_col] = df[code_col].astype(str)
    df[desc_col] = df[desc_col].astype(str)
    # lower-case first, expand second: the order the reference uses (:63)
    df["cleaned_description"] = df[desc_col].map(lambda s: expand_abbreviations(clean_text(s)))
    kept = df[df["cleaned_description"] != ""]
    if len(kept) < len(df):
        print(f"Dropped {len(df) - len(kept)} rows with empty descriptions after cleaning.")
    unique = kept.groupby("cleaned_description").first().reset_index()
    unique = unique.rename(columns={"cleaned_description": "product_description", code_col: "product_code"})
    unique = unique[["product_description", "product_code"]]
    print(f"Found {len(unique)} unique product descriptions for embedding.")
    return unique
