"""TEST INFRASTRUCTURE ONLY: CPU oracle for the EAST hot path (see easa_oracle.c)."""
