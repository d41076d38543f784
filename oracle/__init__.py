"""CPU oracle -- test infrastructure only (see oracle/oracle.py header)."""
