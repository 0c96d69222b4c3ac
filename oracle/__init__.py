"""ORACLE: CPU restatement of the reference's hot path (test infrastructure only)."""
