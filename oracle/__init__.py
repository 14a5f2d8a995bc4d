"""Parity oracle -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms for the mapping hot path.  May be imported only
by tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()``; the product
package ``remixfusion_amd`` never imports it and has no CPU fallback.
"""
