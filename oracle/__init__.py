"""Test infrastructure: CPU restatement of the reference path + reference harness recipe.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
