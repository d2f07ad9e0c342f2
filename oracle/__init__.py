"""CPU restatement of the reference's per-pixel path -- test infrastructure only."""
