"""CPU parity oracle package — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  The product (noaa_apt_amd) never does.  See oracle/apt_oracle.h.
"""
