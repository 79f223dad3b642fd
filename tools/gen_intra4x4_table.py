#!/usr/bin/env python3
"""Generates edge264_amd/csrc/e264_intra_tab.h: the Intra4x4 prediction modes (as the front end resolves them against
neighbour availability, edge264_internal.h:564-634 of the reference) as three taps and a filter type per sample, so that the
kernel predicts a 4x4 block with one table read and three sample reads per lane instead of a 14-way switch.

tap position = byte offset from the block's top-left NEIGHBOUR (row -1, column -1) in the luma tile (32-byte rows):
T(i) -> i + 1, L(j) -> (j + 1) * 32.   entry = off0 | off1 << 8 | off2 << 16 | type << 24,
type = wb | wc << 2 | sh << 3:  value = (a + wb * b + wc * c + ((1 << sh) >> 1)) >> sh.
The generator checks every entry against a direct transcription of the formulas on random neighbours."""
import random

STRIDE = 32
T = lambda i: ("T", i)
L = lambda j: ("L", j)
LP, AVG, COPY = 2 | 4 | 2 << 3, 1 | 1 << 3, 0


def taps(mode, x, y):
    """-> (tap0, tap1, tap2, type) or None for the DC modes"""
    tr = lambda i: T(i) if (i < 4 or mode in (6, 11)) else T(3)
    l_ = lambda j: T(-1) if j < 0 else L(j)
    if mode == 0:
        return T(x), T(x), T(x), COPY
    if mode == 1:
        return L(y), L(y), L(y), COPY
    if mode in (2, 3, 4, 5):
        return None
    if mode in (6, 7):
        if x == 3 and y == 3:
            return tr(6), tr(7), tr(7), LP
        return tr(x + y), tr(x + y + 1), tr(x + y + 2), LP
    if mode == 8:
        if x > y:
            return T(x - y - 2), T(x - y - 1), T(x - y), LP
        if x < y:
            return l_(y - x - 2), L(y - x - 1), L(y - x), LP
        return T(0), T(-1), L(0), LP
    if mode == 9:
        z, i = 2 * x - y, x - (y >> 1)
        if z >= 0 and not z & 1:
            return T(i - 1), T(i), T(i), AVG
        if z >= 0:
            return T(i - 2), T(i - 1), T(i), LP
        if z == -1:
            return L(0), T(-1), T(0), LP
        return L(y - 1), L(y - 2), l_(y - 3), LP
    if mode == 10:
        z, i = 2 * y - x, y - (x >> 1)
        if z >= 0 and not z & 1:
            return l_(i - 1), l_(i), l_(i), AVG
        if z >= 0:
            return l_(i - 2), l_(i - 1), l_(i), LP
        if z == -1:
            return L(0), T(-1), T(0), LP
        return T(x - 1), T(x - 2), T(x - 3), LP
    if mode in (11, 12):
        i = x + (y >> 1)
        if y & 1:
            return tr(i), tr(i + 1), tr(i + 2), LP
        return tr(i), tr(i + 1), tr(i + 1), AVG
    if mode == 13:
        z, i = x + 2 * y, y + (x >> 1)
        if z > 5:
            return L(3), L(3), L(3), COPY
        if z == 5:
            return L(2), L(3), L(3), LP
        if z & 1:
            return L(i), L(i + 1), L(i + 2), LP
        return L(i), L(i + 1), L(i + 1), AVG
    raise ValueError(mode)


def direct(mode, x, y, t, l):
    """the formulas themselves (t: T(-1..7) at index i + 1, l: L(0..3))"""
    Tt = lambda i: t[i + 1]
    lp = lambda a, b, c: (a + 2 * b + c + 2) >> 2
    tr = lambda i: Tt(i) if (i < 4 or mode in (6, 11)) else Tt(3)
    l_ = lambda j: Tt(-1) if j < 0 else l[j]
    if mode == 0: return Tt(x)
    if mode == 1: return l[y]
    if mode in (6, 7):
        return (tr(6) + 3 * tr(7) + 2) >> 2 if (x == 3 and y == 3) else lp(tr(x + y), tr(x + y + 1), tr(x + y + 2))
    if mode == 8:
        if x > y: return lp(Tt(x - y - 2), Tt(x - y - 1), Tt(x - y))
        if x < y: return lp(l_(y - x - 2), l[y - x - 1], l[y - x])
        return lp(Tt(0), Tt(-1), l[0])
    if mode == 9:
        z, i = 2 * x - y, x - (y >> 1)
        if z >= 0 and not z & 1: return (Tt(i - 1) + Tt(i) + 1) >> 1
        if z >= 0: return lp(Tt(i - 2), Tt(i - 1), Tt(i))
        if z == -1: return lp(l[0], Tt(-1), Tt(0))
        return lp(l[y - 1], l[y - 2], l_(y - 3))
    if mode == 10:
        z, i = 2 * y - x, y - (x >> 1)
        if z >= 0 and not z & 1: return (l_(i - 1) + l_(i) + 1) >> 1
        if z >= 0: return lp(l_(i - 2), l_(i - 1), l_(i))
        if z == -1: return lp(l[0], Tt(-1), Tt(0))
        return lp(Tt(x - 1), Tt(x - 2), Tt(x - 3))
    if mode in (11, 12):
        i = x + (y >> 1)
        return lp(tr(i), tr(i + 1), tr(i + 2)) if y & 1 else (tr(i) + tr(i + 1) + 1) >> 1
    if mode == 13:
        z, i = x + 2 * y, y + (x >> 1)
        if z > 5: return l[3]
        if z == 5: return (l[2] + 3 * l[3] + 2) >> 2
        return lp(l[i], l[i + 1], l[i + 2]) if z & 1 else (l[i] + l[i + 1] + 1) >> 1


def off(tap):
    kind, i = tap
    return i + 1 if kind == "T" else (i + 1) * STRIDE


def entry(mode, x, y):
    r = taps(mode, x, y)
    if r is None:
        return 0
    a, b, c, ty = r
    return off(a) | off(b) << 8 | off(c) << 16 | ty << 24


# ---------------------------------------------------------------------------------------------------------------------------------
# Intra8x8: the same trick on the FILTERED edge (8.3.2.2.1), which the kernel lays out in a 32-byte array per wave:
#   FL(j) (left, j = 0..7) at j, the corner FT(-1) = FL(-1) at 8, FT(i) (top and top right, i = 0..15) at 12 + i.
# Nine shapes (c_i8spec of the kernel: the 32 internal modes differ in which neighbours exist, not in the shape):
#   0 vertical, 1 horizontal, 2 DC (no table), 3 diagonal down left, 4 diagonal down right, 5 vertical right, 6 horizontal down,
#   7 vertical left, 8 horizontal up.
FT = lambda i: ("FT", i)
FL = lambda j: ("FT", -1) if j < 0 else ("FL", j)
LP3 = 3 | 2 << 3  # (a + 3 b + 2) >> 2


def taps8(sm, x, y):
    if sm == 0: return FT(x), FT(x), FT(x), COPY
    if sm == 1: return FL(y), FL(y), FL(y), COPY
    if sm == 2: return None
    if sm == 3:
        if x == 7 and y == 7: return FT(14), FT(15), FT(15), LP3
        return FT(x + y), FT(x + y + 1), FT(x + y + 2), LP
    if sm == 4:
        if x > y: return FT(x - y - 2), FT(x - y - 1), FT(x - y), LP
        if x < y: return FL(y - x - 2), FL(y - x - 1), FL(y - x), LP
        return FT(0), FT(-1), FL(0), LP
    if sm == 5:
        z, i = 2 * x - y, x - (y >> 1)
        if z >= 0 and not z & 1: return FT(i - 1), FT(i), FT(i), AVG
        if z >= 0: return FT(i - 2), FT(i - 1), FT(i), LP
        if z == -1: return FL(0), FT(-1), FT(0), LP
        return FL(y - 2 * x - 1), FL(y - 2 * x - 2), FL(y - 2 * x - 3), LP
    if sm == 6:
        z, i = 2 * y - x, y - (x >> 1)
        if z >= 0 and not z & 1: return FL(i - 1), FL(i), FL(i), AVG
        if z >= 0: return FL(i - 2), FL(i - 1), FL(i), LP
        if z == -1: return FL(0), FT(-1), FT(0), LP
        return FT(x - 2 * y - 1), FT(x - 2 * y - 2), FT(x - 2 * y - 3), LP
    if sm == 7:
        i = x + (y >> 1)
        if y & 1: return FT(i), FT(i + 1), FT(i + 2), LP
        return FT(i), FT(i + 1), FT(i + 1), AVG
    if sm == 8:
        z, i = x + 2 * y, y + (x >> 1)
        if z > 13: return FL(7), FL(7), FL(7), COPY
        if z == 13: return FL(6), FL(7), FL(7), LP3
        if z & 1: return FL(i), FL(i + 1), FL(i + 2), LP
        return FL(i), FL(i + 1), FL(i + 1), AVG
    raise ValueError(sm)


def direct8(sm, x, y, ft, fl):
    """8.3.2.2.2-9 on the filtered edge (ft: FT(-1..15) at index i + 1, fl: FL(0..7)), written out independently of taps8"""
    T_ = lambda i: ft[i + 1]
    L_ = lambda j: ft[0] if j < 0 else fl[j]
    lp = lambda a, b, c: (a + 2 * b + c + 2) >> 2
    if sm == 0: return T_(x)
    if sm == 1: return fl[y]
    if sm == 3: return (T_(14) + 3 * T_(15) + 2) >> 2 if (x == 7 and y == 7) else lp(T_(x + y), T_(x + y + 1), T_(x + y + 2))
    if sm == 4:
        if x > y: return lp(T_(x - y - 2), T_(x - y - 1), T_(x - y))
        if x < y: return lp(L_(y - x - 2), L_(y - x - 1), L_(y - x))
        return lp(T_(0), T_(-1), fl[0])
    if sm == 5:
        z, i = 2 * x - y, x - (y >> 1)
        if z >= 0 and z % 2 == 0: return (T_(i - 1) + T_(i) + 1) >> 1
        if z >= 0: return lp(T_(i - 2), T_(i - 1), T_(i))
        if z == -1: return lp(fl[0], T_(-1), T_(0))
        return lp(L_(y - 2 * x - 1), L_(y - 2 * x - 2), L_(y - 2 * x - 3))
    if sm == 6:
        z, i = 2 * y - x, y - (x >> 1)
        if z >= 0 and z % 2 == 0: return (L_(i - 1) + L_(i) + 1) >> 1
        if z >= 0: return lp(L_(i - 2), L_(i - 1), L_(i))
        if z == -1: return lp(fl[0], T_(-1), T_(0))
        return lp(T_(x - 2 * y - 1), T_(x - 2 * y - 2), T_(x - 2 * y - 3))
    if sm == 7:
        i = x + (y >> 1)
        return lp(T_(i), T_(i + 1), T_(i + 2)) if y & 1 else (T_(i) + T_(i + 1) + 1) >> 1
    if sm == 8:
        z, i = x + 2 * y, y + (x >> 1)
        if z > 13: return fl[7]
        if z == 13: return (fl[6] + 3 * fl[7] + 2) >> 2
        return lp(fl[i], fl[i + 1], fl[i + 2]) if z & 1 else (fl[i] + fl[i + 1] + 1) >> 1


def off8(tap):
    kind, i = tap
    return i if kind == "FL" else (8 if i < 0 else 12 + i)


def table8(rnd):
    rows = []
    for sm in range(9):
        row = []
        for p in range(64):
            x, y = p & 7, p >> 3
            r = taps8(sm, x, y)
            if r is None:
                row.append(0)
                continue
            a, b, c, ty = r
            e = off8(a) | off8(b) << 8 | off8(c) << 16 | ty << 24
            row.append(e)
            for _ in range(20):
                ft = [rnd.randrange(256) for _ in range(17)]
                fl = [rnd.randrange(256) for _ in range(8)]
                fz = {j: fl[j] for j in range(8)}
                fz[8] = ft[0]
                for i in range(16): fz[12 + i] = ft[i + 1]
                va, vb, vc, sh = fz[e & 255], fz[e >> 8 & 255], fz[e >> 16 & 255], ty >> 3
                v = (va + (ty & 3) * vb + (vc if ty & 4 else 0) + ((1 << sh) >> 1)) >> sh
                assert v == direct8(sm, x, y, ft, fl), (sm, x, y)
        rows.append(row)
    return rows


def main():
    rnd = random.Random(1)
    rows = []
    for mode in range(14):
        row = []
        for p in range(16):
            x, y = p & 3, p >> 2
            e = entry(mode, x, y)
            row.append(e)
            if taps(mode, x, y) is None:
                continue
            for _ in range(50):
                t = [rnd.randrange(256) for _ in range(9)]
                l = [rnd.randrange(256) for _ in range(4)]
                tile = {}
                for i in range(-1, 8): tile[i + 1] = t[i + 1]
                for j in range(4): tile[(j + 1) * STRIDE] = l[j]
                a, b, c, ty = tile[e & 255], tile[e >> 8 & 255], tile[e >> 16 & 255], e >> 24
                sh = ty >> 3
                v = (a + (ty & 3) * b + (c if ty & 4 else 0) + ((1 << sh) >> 1)) >> sh
                assert v == direct(mode, x, y, t, l), (mode, x, y)
        rows.append(row)
    print("// GENERATED by tools/gen_intra4x4_table.py -- do not edit.  Intra4x4 modes as taps: see the generator's header.")
    print("__constant__ uint32_t c_i4tab[14 * 16] = {")
    for mode, row in enumerate(rows):
        print("\t" + ", ".join(f"0x{e:08x}u" for e in row) + f", // mode {mode}")
    print("};")
    print("// Intra8x8 shapes (c_i8spec) as taps on the filtered edge array of the wave: FL(j) at j, the corner at 8, FT(i) at 12 + i.")
    print("__constant__ uint32_t c_i8tab[9 * 64] = {")
    for sm, row in enumerate(table8(rnd)):
        for y in range(8):
            print("\t" + ", ".join(f"0x{e:08x}u" for e in row[8 * y:8 * y + 8]) + "," + (f" // shape {sm}" if y == 0 else ""))
    print("};")


if __name__ == "__main__":
    main()
