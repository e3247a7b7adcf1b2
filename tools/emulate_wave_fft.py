"""CPU emulator of the wave-per-frame-pair STFT kernel's index math (tools only; no GPU here).

Mirrors kernels_wave.hip lane by lane: 64 lanes x P points, passes [R1, R2, R3], LDS exchange layouts with
padding, the adjacent-pair butterfly assignment of the last pass, the Hermitian untangle through partner
lanes, and reports LDS bank conflicts of every access stream (per MI355X_MICROARCH.md LDS table).
"""
import sys
import numpy as np


def dft(u, R, sgn=-1):
    t = np.arange(R)
    return np.array([np.sum(u * np.exp(sgn * 2j * np.pi * t * r / R)) for r in range(R)])


def conflicts(addr_elems, kind):
    """addr_elems: per-lane element index (8-byte complex units) for ONE wave instruction.
    kind: 'w64' ds_write_b64 (16-lane contiguous groups, bank = (a/4)%32), 'r64' ds_read_b64 (2x32, (a/4)%64),
    'r128' ds_read_b128 (4 groups of 16 as listed in the guide, (a/4)%64), 'w128' (8-lane groups, %32)."""
    a = np.asarray(addr_elems) * 8
    if kind == 'w64':
        groups = [range(g * 16, g * 16 + 16) for g in range(4)]; mod = 32; width = 2
    elif kind == 'r64':
        groups = [range(0, 32), range(32, 64)]; mod = 64; width = 2
    elif kind == 'r128':
        groups = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
                  [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59], [36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
        mod = 64; width = 4
    elif kind == 'w128':
        groups = [range(g * 8, g * 8 + 8) for g in range(8)]; mod = 32; width = 4
    worst = 1
    for g in groups:
        banks = {}
        for l in g:
            for d in range(width):
                b = (a[l] // 4 + d) % mod
                banks.setdefault(b, set()).add(a[l] // 4 + d)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def run(K=1024, verbose=True):
    P = K // 64
    assert K in (1024, 2048)
    R1, R2, R3 = (16, 16, K // 256)
    rng = np.random.default_rng(0)
    xa, xb = rng.standard_normal(K), rng.standard_normal(K)
    x = xa + 1j * xb
    lanes = np.arange(64)
    worst = {}
    def note(name, addrs, kind):
        worst[name] = max(worst.get(name, 1), conflicts(addrs, kind))

    # ---- pass A: radix R1, p = 1.  q1 = K/R1 butterflies, B1 = q1/64 per lane: i = l + 64 u
    q1 = K // R1; B1 = q1 // 64
    pad1 = lambda e: e + (e >> 4)              # one complex of padding every 16
    lds = np.zeros(K + K // 16 + 64, complex)
    for u in range(B1):
        for l in lanes:
            i = l + 64 * u
            uu = np.array([x[i + t * q1] for t in range(R1)])
            v = dft(uu, R1)
            for r in range(R1):
                lds[pad1(R1 * i + r)] = v[r]
        for r in range(R1):
            note('A.write', [pad1(R1 * (l + 64 * u) + r) for l in lanes], 'w64')
    # ---- pass B: radix R2, p = R1
    p = R1; q2 = K // R2; B2 = q2 // 64
    regs = {}
    for u in range(B2):
        for l in lanes:
            i = l + 64 * u; k = i & (p - 1)
            uu = np.array([lds[pad1(i + t * q2)] * np.exp(-2j * np.pi * t * k / (p * R2)) for t in range(R2)])
            regs[(l, u)] = (dft(uu, R2), i, k)
        for t in range(R2):
            note('B.read', [pad1(l + 64 * u + t * q2) for l in lanes], 'r64')
    lds2 = np.zeros(K, complex)
    for (l, u), (v, i, k) in regs.items():
        for r in range(R2):
            lds2[(i - k) * R2 + k + r * p] = v[r]
    for u in range(B2):
        for r in range(R2):
            note('B.write', [((l + 64 * u) - ((l + 64 * u) & (p - 1))) * R2 + ((l + 64 * u) & (p - 1)) + r * p for l in lanes], 'w64')
    # ---- pass C: radix R3, p = R1*R2 = 256; q3 = K/R3 butterflies, adjacent pairs: i = 2 l + e + 128 u
    p = R1 * R2; q3 = K // R3; B3 = q3 // 64      # B3 = 4 (K=1024,R3=4) or 4 (K=2048,R3=8)
    Z = {}
    for u in range(B3 // 2):
        for l in lanes:
            for e in range(2):
                i = 2 * l + e + 128 * u; k = i & (p - 1)
                uu = np.array([lds2[i + t * q3] * np.exp(-2j * np.pi * t * k / (p * R3)) for t in range(R3)])
                v = dft(uu, R3)
                j = (i - k) * R3 + k
                for r in range(R3):
                    Z[(l, e, u, r)] = (j + r * p, v[r])
        for t in range(R3):
            note('C.read128', [2 * l + 128 * u + t * q3 for l in lanes], 'r128')
    # check the complex FFT and the lane->bin map: lane holds bins 2l + e + 128 q
    ref = np.fft.fft(x)
    err = max(abs(v - ref[kk]) for (kk, v) in Z.values())
    nq = K // 128
    for (l, e, u, r), (kk, v) in Z.items():
        q = (kk - 2 * l - e) // 128
        assert kk == 2 * l + e + 128 * q and 0 <= q < nq, (l, e, u, r, kk)
    # ---- untangle through partner lanes
    byidx = {}
    for (l, e, u, r), (kk, v) in Z.items():
        q = (kk - 2 * l - e) // 128
        byidx[(l, e, q)] = v
    XA = np.zeros(K, complex); XB = np.zeros(K, complex)
    for l in lanes:
        for q in range(nq):
            for e in range(2):
                if e == 0:
                    pl = (64 - l) & 63
                    pq = (nq - 1 - q) if l != 0 else ((nq - q) % nq)
                else:
                    pl = 63 - l; pq = nq - 1 - q
                zk = byidx[(l, e, q)]; zp = byidx[(pl, e, pq)]
                k = 2 * l + e + 128 * q
                assert (K - k) % K == 2 * pl + e + 128 * pq, (l, e, q)
                XA[k] = 0.5 * (zk + np.conj(zp)); XB[k] = -0.5j * (zk - np.conj(zp))
    ea = np.abs(XA - np.fft.fft(xa)).max(); eb = np.abs(XB - np.fft.fft(xb)).max()
    if verbose:
        print(f"K={K} radices {R1},{R2},{R3}: complex fft err {err:.2e}, untangle err A {ea:.2e} B {eb:.2e}")
        print("  worst-case LDS conflict ways per access stream:", worst)
    assert err < 1e-9 and ea < 1e-9 and eb < 1e-9
    return worst


if __name__ == "__main__":
    for K in (1024, 2048):
        run(K)
