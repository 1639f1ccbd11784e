"""CPU oracle (test infrastructure only): see oracle/refcpu.py."""
