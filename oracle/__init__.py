"""CPU restatement of the reference's algorithms.  Test infrastructure only: imported by tests/, by
__graft_entry__.smoke() and by the cpu_baseline leg of bench.py, never by the product path."""
