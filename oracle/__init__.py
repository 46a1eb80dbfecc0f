"""CPU oracle (test infrastructure only -- see cae_oracle.py header)."""
