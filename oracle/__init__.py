"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference kriging execute path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (pykrige_amd) never does.
"""
