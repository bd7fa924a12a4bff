"""CPU suite, only where the reference checkout is present (this container): the data
pipeline and the best-match bookkeeping checked against the reference's own DATA FILES
and RECORDED OUTPUT (`analysis_results/best_usda_matches.{csv,xlsx}`).  Data is read as
data; no reference code is imported or run.  Skipped on the GPU box, which has no
/root/reference."""
import os

import numpy as np
import pandas as pd
import pytest

REF = os.environ.get("VECTORDB_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "data", "Actuals", "Transaction_Report_Actual.xlsx")),
                                reason="reference data files not present")


@pytest.fixture(scope="module")
def ref_cfg():
    from vectordb_amd import config
    saved = {k: getattr(config, k) for k in ("TRANSACTION_REPORT_FILE", "GROUND_TRUTH_FILE")}
    config.TRANSACTION_REPORT_FILE = os.path.join(REF, "data", "Actuals", "Transaction_Report_Actual.xlsx")
    config.GROUND_TRUTH_FILE = os.path.join(REF, "data", "Actuals", "Corrected_mapping.xlsx")
    yield config
    for k, v in saved.items():
        setattr(config, k, v)


@pytest.fixture(scope="module")
def recorded():
    return pd.read_csv(os.path.join(REF, "analysis_results", "best_usda_matches.csv"))


def test_xlsx_reader_equals_the_csv_twin(recorded):
    """The stdlib workbook reader on the reference's own .xlsx == pandas on its .csv twin."""
    from vectordb_amd.data_processing import read_excel
    x = read_excel(os.path.join(REF, "analysis_results", "best_usda_matches.xlsx"))
    assert list(x.columns) == list(recorded.columns) and len(x) == len(recorded) == 15235
    for col in x.columns:
        if recorded[col].dtype.kind == "f":
            assert np.allclose(x[col].astype(float), recorded[col], rtol=1e-12, equal_nan=True), col
        else:
            assert (x[col].astype(str) == recorded[col].astype(str)).all(), col


def test_pipeline_on_the_real_workbooks(ref_cfg, capsys):
    from vectordb_amd import data_processing as dp
    from vectordb_amd.vectordb import build_usda_lookup
    raw = dp.load_transaction_data()
    assert raw.shape == (17034, 21) and raw["ProductCode"].dtype == object
    # two descriptions are the literal text "None": missing values to pandas' reader, hence to this one
    assert int(raw["ProductDescription"].isna().sum()) == 3 and not (raw["ProductDescription"].astype(str) == "None").any()
    uniq = dp.process_transaction_data(raw)
    assert list(uniq.columns) == ["product_description", "product_code"] and len(uniq) == 14777   # SURVEY's C1 size
    assert uniq["product_description"].is_monotonic_increasing and uniq["product_description"].is_unique
    assert (uniq["product_description"] == uniq["product_description"].str.lower()).all()
    lookup = build_usda_lookup()
    assert len(lookup) == 149 and lookup["040948"] == "Ground beef 73%"      # '1040948' minus its leading digit


def test_best_match_rows_and_ground_truth_columns_equal_the_recorded_output(ref_cfg, recorded, oracle_lib, tmp_path):
    """generate_best_usda_matches on the real transactions (hashed 64-d embeddings: the model is not
    available, so similarity scores and best codes are NOT comparable) must produce the recorded
    output's rows in its order, with its product_code / product_description / known_usda_code columns,
    one-element id lists, and is_correct_match consistent with known == best."""
    from vectordb_amd import analysis, data_processing as dp
    from vectordb_amd.embedders import HashEmbedder
    from vectordb_amd.vectordb import ProductVectorDB
    raw = dp.load_transaction_data()
    uniq = dp.process_transaction_data(raw)
    db = ProductVectorDB(persist_directory=str(tmp_path / "s"), embed_fn=HashEmbedder(64), usda_lookup_map={}, lib=oracle_lib)
    db.add_products_to_db(uniq, recreate=True)
    assert db.collection.count() == 14777
    mapping_df = analysis.load_ground_truth_mapping()
    usda_codes = analysis.get_unique_usda_codes(mapping_df)
    assert len(usda_codes) == recorded["best_matching_usda_code"].nunique() == 37
    p2u = analysis.create_product_to_usda_mapping(raw, mapping_df)
    usda_emb = {c: db.embedder.embed_query(c) for c in usda_codes}
    out = analysis.generate_best_usda_matches(db, mapping_df, raw, usda_emb, p2u)
    assert list(out.columns) == list(recorded.columns)
    assert len(out) == len(recorded)
    assert (out["product_code"].astype(str).values == recorded["product_code"].astype(str).values).all()
    assert (out["product_description"].astype(str).values == recorded["product_description"].astype(str).values).all()
    rec_known = recorded["known_usda_code"].where(recorded["known_usda_code"].notna(), None)
    assert [k if k is not None else None for k in out["known_usda_code"]] == list(rec_known)
    assert all(isinstance(i, list) and len(i) == 1 for i in out["product_id"])          # recorded: "['item_9655']"
    assert all(s.startswith("['item_") for s in recorded["product_id"])
    known = out["known_usda_code"].notna()
    assert int(known.sum()) == int(recorded["is_correct_match"].notna().sum()) == 73
    assert all(bool(r["is_correct_match"]) == (r["known_usda_code"] == r["best_matching_usda_code"]) for _, r in out[known].iterrows())
    # and the recorded output obeys the same rule
    rk = recorded[recorded["known_usda_code"].notna()]
    assert ((rk["is_correct_match"].astype(str) == "True") == (rk["known_usda_code"] == rk["best_matching_usda_code"])).all()


def test_summary_counts_of_the_recorded_output(recorded):
    """The counts generate_best_usda_matches.py prints at its end (:291-325), computed by this build's
    summarize_best_matches on the reference's recorded output."""
    from vectordb_amd import analysis
    s = analysis.summarize_best_matches(recorded)
    assert s["total_products"] == 15235 and s["known"] == 73 and s["unknown"] == 15162
    assert s["correct"] == 31 and s["incorrect"] == 42 and s["accuracy"] == pytest.approx(31 / 73)
