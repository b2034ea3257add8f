"""CPU oracles (test infrastructure only -- see the header of each module)."""
