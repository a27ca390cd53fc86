#!/usr/bin/env python3
"""The table of rc_refexp.h: for k = 0..127, T = nearest double of 2^(k/128) / H - 1 and H = nearest double of 2^(k/128), stored as the bits of T
and the bits of H minus (k << 52) / 128 (the layout of glibc's / ARM optimized-routines' __exp_data.tab).  python tools/make_exp_table.py"""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 100
ln2 = Decimal(2).ln()


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


vals = []
for k in range(128):
    v = (Decimal(k) / 128 * ln2).exp()
    h = float(v)
    t = float(v / Decimal(h) - 1)
    vals += [bits(t), bits(h) - ((k << 52) // 128)]
for i in range(0, 256, 4):
    print("  " + " ".join("0x%016xull," % v for v in vals[i:i + 4]))
