"""CPU oracle: restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY (see oracle/README.md)."""
