"""TEST INFRASTRUCTURE ONLY: CPU parity checker for the HIP path (see rfops_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product packages (rfnet_amd, tf_ops, pc_distance) never do.
"""
