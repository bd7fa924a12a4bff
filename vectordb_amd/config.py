"""Constants mirrored from the reference's src/config.py:8-59 so that scripts
written against it import the same names.  (No print on import, unlike
src/config.py:61.)"""
import os
from pathlib import Path

PROJECT_ROOT = Path(os.environ.get("VECTORDB_PROJECT_ROOT", Path.cwd()))

DATA_DIR = PROJECT_ROOT / "data"
ACTUALS_DATA_DIR = DATA_DIR / "Actuals"

TRANSACTION_REPORT_FILE = ACTUALS_DATA_DIR / "Transaction_Report_Actual.xlsx"
TRANSACTION_SHEET_NAME = "Sheet1"
TRANSACTION_PRODUCT_CODE_COL = "ProductCode"
TRANSACTION_DESC_COL = "ProductDescription"

GROUND_TRUTH_FILE = ACTUALS_DATA_DIR / "Corrected_mapping.xlsx"
GROUND_TRUTH_SHEET_NAME = "Sheet1"
GROUND_TRUTH_ID_COLS = ["Fulton_code", "Pritzlaff_code", "Queen_code", "Moesle_code", "Anmar_code"]
GROUND_TRUTH_USDA_COL = "USDA_Code"

EMBEDDING_MODEL = "all-mpnet-base-v2"
EMBEDDING_DIM = 768

CHROMA_DB_PATH = PROJECT_ROOT / "chroma_db_actuals"
COLLECTION_NAME = "actual_products"

N_RESULTS_INITIAL_SEARCH = 30
N_RESULTS_FINAL = 5
SIMILARITY_THRESHOLD_FORWARD = 0.25
SIMILARITY_THRESHOLD_BACKWARD = 0.20

MIN_TRANSACTION_COUNT = 1

# engine settings (no reference counterpart)
STORE_DTYPE = os.environ.get("VECTORDB_STORE_DTYPE", "f32")   # "f32" | "bf16"
DEVICE = int(os.environ.get("VECTORDB_DEVICE", "0"))
