"""ORACLE - test infrastructure only (see nrv_oracle.py / nrv_oracle.c headers)."""
