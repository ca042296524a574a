"""ORACLE — test infrastructure only (see the header of each module).

CPU restatement of the reference's sampling hot path.  Nothing under surfd_amd/ imports
this package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
"""
