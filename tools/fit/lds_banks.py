#!/usr/bin/env python3
"""dev tool: LDS bank-conflict check of the wave-per-frame FFT's exchange layouts (csrc/stft.hip, wave_fft1024).

A ds_read_b64 / ds_write_b64 wave instruction is served in two halves of 32 lanes; a half is conflict-free when its 32 eight-byte
words fall into 32 distinct word-pairs of the 64 four-byte banks, i.e. (word index mod 32) is distinct across the half.  Prints the
worst multiplicity per access pattern (1 = conflict-free)."""
import numpy as np


def worst(addr_fn, nregs):
    w = 0
    for r in range(nregs):
        for half in (0, 1):
            lanes = np.arange(32) + 32 * half
            words = np.array([addr_fn(int(l), r) for l in lanes])
            _, cnt = np.unique(words % 32, return_counts=True)
            # identical addresses broadcast: count distinct addresses per bank
            mult = 0
            for b in np.unique(words % 32):
                mult = max(mult, len(np.unique(words[words % 32 == b])))
            w = max(w, mult)
    return w


S1, T = 68, 264
pats = {
    "input read  n = l + 64 r (natural)": (lambda l, r: l + 64 * r, 16),
    "ex1 write A1(k1 = r, l)": (lambda l, r: S1 * r + l, 16),
    "ex1 read  A1(k1 = l >> 2, (l & 3) + 4 r)": (lambda l, r: S1 * (l >> 2) + (l & 3) + 4 * r, 16),
    "ex2 write A2(k1 = l >> 2, k2 = r, j = l & 3)": (lambda l, r: T * (l & 3) + (l >> 2) + 16 * r, 16),
    "ex2 read  A2(k1 = l & 15, k2 = 4 q + (l >> 4), j), r = 4 q + j": (lambda l, r: T * (r & 3) + (l & 15) + 16 * (4 * (r >> 2) + (l >> 4)), 16),
    "final write k = k1 + 16 k2 + 256 k3, r = 4 q + k3": (lambda l, r: (l & 15) + 16 * (4 * (r >> 2) + (l >> 4)) + 256 * (r & 3), 16),
    "T1 read [k1 = r][l]": (lambda l, r: 64 * r + l, 16),
    "T2 read [k2 = r][j = l & 3]": (lambda l, r: 4 * r + (l & 3), 16),
    "split read k = 1 + l + 64 i": (lambda l, r: 1 + l + 64 * r, 8),
    "split read M - k": (lambda l, r: 1024 - (1 + l + 64 * r), 8),
}
for name, (fn, n) in pats.items():
    print(f"{worst(fn, n)}-way  {name}")
# the 512-point transform (n_fft = 1024): 8 points per lane, 8 x 8 x 8, pads 72 / 68, output in natural order
pats8 = {
    "P8 input read n = l + 64 r": (lambda l, r: l + 64 * r, 8),
    "P8 ex1 write 72 k1 + l": (lambda l, r: 72 * r + l, 8),
    "P8 ex1 read  72 (l >> 3) + (l & 7) + 8 r": (lambda l, r: 72 * (l >> 3) + (l & 7) + 8 * r, 8),
    "P8 ex2 write 68 (l & 7) + (l >> 3) + 8 k2": (lambda l, r: 68 * (l & 7) + (l >> 3) + 8 * r, 8),
    "P8 ex2 read  68 j + l": (lambda l, r: 68 * r + l, 8),
    "P8 T1 read [k1][l]": (lambda l, r: 64 * r + l, 8),
    "P8 T2 read [k2][j = l & 7]": (lambda l, r: 8 * r + (l & 7), 8),
}
for name, (fn, n) in pats8.items():
    print(f"{worst(fn, n)}-way  {name}")
