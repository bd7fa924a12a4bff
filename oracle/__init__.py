"""CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/vdb_oracle.c header)."""
