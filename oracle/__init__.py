"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/sgo_oracle.c header)."""
