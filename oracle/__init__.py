"""CPU oracle for the FDTD hot path -- test infrastructure, never shipped.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
