"""Thin re-export so tests can `import oracle_binding` (the checker lives in oracle/binding.py)."""
from oracle.binding import *  # noqa: F401,F403
from oracle.binding import load_oracle, load_ref, FASTQ, FASTA, Oracle, RefLib  # noqa: F401
