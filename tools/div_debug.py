#!/usr/bin/env python3
"""Where the table-reciprocal division by a count differs from the division (lab library): mismatches by dividend exponent range."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
lab = ge.load_lab(); lab.set_device(0)
for n in (10, 1000):
    for lo, hi in ((0, 0x00800000), (0x00800000, 0x01000000), (0x01000000, 0x02800000), (0x02800000, 0x0D000000), (0x0D000000, 0x7F800000),
                   (0x7F800000, 0x80000000), (0x80000000, 0x80800000), (0x80800000, 0x82800000), (0x82800000, 0xFF800000), (0xFF800000, 0x100000000)):
        bad, ex, exn = lab.div_compare(n, 1, lo, hi - lo)
        print(n, hex(lo), hex(hi), bad, hex(ex), flush=True)
