"""ORACLE -- test infrastructure only (see oracle/orc_common.h).  PARITY UNPINNED.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
