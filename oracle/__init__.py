"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- see oracle/oracle.h.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product package `circl_amd` never does.
"""
