"""Parity oracle package — TEST INFRASTRUCTURE (see oracle/dvm_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
