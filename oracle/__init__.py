"""CPU oracle of the reference hot path: test infrastructure only (see oracle/README.md)."""
