"""Tolerance-aware top-k comparison (BASELINE.json: scores within 1e-5 relative,
identical id lists except for ties inside that tolerance)."""
import numpy as np

REL_TOL = 1e-5      # north_star tolerance on scores
ABS_FLOOR = 2e-6    # fp32 rounding of 1 - s near s ~ 1


def exact_scores(rows_norm, qn, ids):
    """fp64 cosine of normalised query qn against the given rows."""
    return rows_norm[ids].astype(np.float64) @ qn.astype(np.float64)


def check_topk(idx, dist, ref_idx, ref_dist, ref_score, rows_norm, queries_norm, row_offset=0, what=""):
    """Returns the number of positions whose id differs from the oracle (all of
    which must be in-tolerance ties)."""
    b, k = ref_idx.shape
    assert idx.shape == (b, k) and dist.shape == (b, k), f"{what}: shape {idx.shape} vs {(b, k)}"
    swaps = 0
    for q in range(b):
        valid = ref_idx[q] >= 0
        nv = int(valid.sum())
        assert np.all(idx[q, nv:] == -1) and np.all(np.isinf(dist[q, nv:])), f"{what}: q{q} padding"
        got = idx[q, :nv] - row_offset
        assert len(set(got.tolist())) == nv and got.min(initial=0) >= 0, f"{what}: q{q} duplicate/invalid ids {got}"
        tol = REL_TOL * np.maximum(np.abs(ref_score[q, :nv]), 1e-3) + ABS_FLOOR
        # distances agree position by position
        assert np.all(np.abs(dist[q, :nv].astype(np.float64) - ref_dist[q, :nv]) <= tol), \
            f"{what}: q{q} distances differ: {dist[q, :nv]} vs {ref_dist[q, :nv]}"
        # returned distance is the true distance of the returned row
        true_s = exact_scores(rows_norm, queries_norm[q], got)
        assert np.all(np.abs((1.0 - true_s) - dist[q, :nv]) <= tol), f"{what}: q{q} distance != 1-cos of returned row"
        # ascending
        assert np.all(np.diff(dist[q, :nv].astype(np.float64)) >= -tol[1:] if nv > 1 else True), f"{what}: q{q} not ascending"
        diff = got != ref_idx[q, :nv]
        if diff.any():
            # every mismatching row must tie (within tolerance) with the oracle's row at that rank
            assert np.all(np.abs(true_s[diff] - ref_score[q, :nv][diff]) <= tol[diff]), \
                f"{what}: q{q} ids differ outside tolerance: {got} vs {ref_idx[q, :nv]}"
            swaps += int(diff.sum())
    return swaps


def recall_at_k(idx, ref_idx):
    hits = 0
    total = 0
    for q in range(ref_idx.shape[0]):
        ref = set(int(i) for i in ref_idx[q] if i >= 0)
        hits += len(ref & set(int(i) for i in idx[q]))
        total += len(ref)
    return hits / max(total, 1)


def write_xlsx(path, header, rows, sheet_name="Sheet1", inline_from_col=None):
    """Minimal workbook writer for fixtures: str cells as shared strings (or inline
    strings from column `inline_from_col` on), int/float as numbers, bool as 'b',
    None as an absent cell."""
    import zipfile
    from xml.sax.saxutils import escape

    shared, shared_ix = [], {}

    def col_name(j):
        s = ""
        j += 1
        while j:
            j, r = divmod(j - 1, 26)
            s = chr(65 + r) + s
        return s

    def cell(ref, j, v):
        if v is None:
            return ""
        if isinstance(v, bool):
            return f'<c r="{ref}" t="b"><v>{int(v)}</v></c>'
        if isinstance(v, (int, float)):
            return f'<c r="{ref}"><v>{v!r}</v></c>'
        if inline_from_col is not None and j >= inline_from_col:
            return f'<c r="{ref}" t="inlineStr"><is><t xml:space="preserve">{escape(v)}</t></is></c>'
        if v not in shared_ix:
            shared_ix[v] = len(shared)
            shared.append(v)
        return f'<c r="{ref}" t="s"><v>{shared_ix[v]}</v></c>'

    ns = 'xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main"'
    body = []
    for i, r in enumerate([list(header)] + [list(r) for r in rows]):
        cells = "".join(cell(f"{col_name(j)}{i + 1}", j, v) for j, v in enumerate(r))
        if cells:
            body.append(f'<row r="{i + 1}">{cells}</row>')
    sheet = f'<?xml version="1.0" encoding="UTF-8"?><worksheet {ns}><sheetData>{"".join(body)}</sheetData></worksheet>'
    sst = (f'<?xml version="1.0" encoding="UTF-8"?><sst {ns} count="{len(shared)}" uniqueCount="{len(shared)}">'
           + "".join(f'<si><t xml:space="preserve">{escape(s)}</t></si>' for s in shared) + "</sst>")
    wb = (f'<?xml version="1.0" encoding="UTF-8"?><workbook {ns} '
          'xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"><sheets>'
          f'<sheet name="{escape(sheet_name)}" sheetId="1" r:id="rId1"/></sheets></workbook>')
    rels = ('<?xml version="1.0" encoding="UTF-8"?>'
            '<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
            '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" '
            'Target="worksheets/sheet1.xml"/>'
            '<Relationship Id="rId2" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/sharedStrings" '
            'Target="sharedStrings.xml"/></Relationships>')
    ctypes_ = ('<?xml version="1.0" encoding="UTF-8"?><Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">'
               '<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/>'
               '<Default Extension="xml" ContentType="application/xml"/></Types>')
    root_rels = ('<?xml version="1.0" encoding="UTF-8"?>'
                 '<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                 '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/officeDocument" '
                 'Target="xl/workbook.xml"/></Relationships>')
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("[Content_Types].xml", ctypes_)
        z.writestr("_rels/.rels", root_rels)
        z.writestr("xl/workbook.xml", wb)
        z.writestr("xl/_rels/workbook.xml.rels", rels)
        z.writestr("xl/sharedStrings.xml", sst)
        z.writestr("xl/worksheets/sheet1.xml", sheet)


def record_gate(name: str, text: str) -> None:
    """Print a full-size gate's figures AND leave them behind in files (pytest -q hides prints): appended to
    profiles/gates_last_run.txt and, where a gpurun scratch directory exists, to gpurun_out/gates_last_run.txt."""
    import os
    import time
    line = f"[{time.strftime('%Y-%m-%d %H:%M:%S')}] {name}: {text}"
    print("\n" + line)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in ("profiles", "gpurun_out"):
        try:
            if os.path.isdir(os.path.join(root, d)):
                with open(os.path.join(root, d, "gates_last_run.txt"), "a") as fh:
                    fh.write(line + "\n")
        except OSError:
            pass
