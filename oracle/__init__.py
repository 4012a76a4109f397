"""CPU oracle for the Atlas retrieval hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
atlas_amd/ (the product) never imports it.
"""
