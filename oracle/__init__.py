"""CPU oracle of the hot path — TEST INFRASTRUCTURE, never imported by capreolus_amd/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(as the checker / the timed CPU baseline).  Parity pinning: tests/test_oracle_golden.py checks
every function here against golden vectors produced by the reference nn.Modules.
"""
