"""Synthetic workload generator.

Bit-compatible with ``std::mt19937`` so the same seeds give the same sequences in
the C++ drivers, the reference harness (oracle/ref_harness.cpp ``synth``) and Python:
residue = "ARNDCQEGHILKMFPSTWYV"[g() % 20], query drawn first, then template.
"""
import numpy as np

AA20 = "ARNDCQEGHILKMFPSTWYV"
_AA20 = np.frombuffer(AA20.encode(), dtype=np.uint8)


class MT19937:
    """std::mt19937 (init_genrand seeding, 32-bit tempering), block-vectorised."""

    N, M = 624, 397

    def __init__(self, seed):
        mt = np.empty(self.N, dtype=np.uint64)
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, self.N):
            mt[i] = (1812433253 * (int(mt[i - 1]) ^ (int(mt[i - 1]) >> 30)) + i) & 0xFFFFFFFF
        self.mt = mt.astype(np.uint32)
        self.buf = np.empty(0, dtype=np.uint32)

    def _twist(self):
        mt = self.mt
        N, M = self.N, self.M
        UP, LO = np.uint32(0x80000000), np.uint32(0x7FFFFFFF)
        MAG = np.uint32(0x9908B0DF)

        def step(lo, hi):
            # new mt[k] for k in [lo,hi) depends on mt[k], mt[k+1], mt[(k+M)%N]; valid while sources are old/new as in the scalar loop
            k = np.arange(lo, hi)
            y = (mt[k] & UP) | (mt[(k + 1) % N] & LO)
            mt[k] = mt[(k + M) % N] ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), MAG, np.uint32(0))

        # k in [0, N-M): sources k+M are old values -> one vector op is exact
        step(0, N - M)
        # k in [N-M, N-1): sources (k+M)%N = k-(N-M) are NEW values computed above, in blocks of N-M
        lo = N - M
        while lo < N - 1:
            hi = min(lo + (N - M), N - 1)
            step(lo, hi)
            lo = hi
        step(N - 1, N)

    def draw(self, n):
        if n <= 0:
            return np.empty(0, dtype=np.uint32)
        out = []
        have = 0
        if len(self.buf):
            out.append(self.buf)
            have = len(self.buf)
        while have < n:
            self._twist()
            y = self.mt.copy()
            y ^= y >> np.uint32(11)
            y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
            y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
            y ^= y >> np.uint32(18)
            out.append(y)
            have += len(y)
        allv = np.concatenate(out)
        self.buf = allv[n:]
        return allv[:n]


def residues(g, n):
    """n residues as a str, A[g() % 20] each."""
    return _AA20[(g.draw(n) % np.uint32(20)).astype(np.int64)].tobytes().decode()


def random_pair(seed, qlen, tlen=None):
    g = MT19937(seed)
    q = residues(g, qlen)
    t = residues(g, qlen if tlen is None else tlen)
    return q, t


def homolog_pair(seed, n, sub_rate=0.15, indel=5):
    """Query random; template = copy with `sub_rate` substitutions, one `indel`-residue
    deletion and one insertion (keeps length n) so tracebacks are long."""
    g = MT19937(seed)
    q = residues(g, n)
    t = list(q)
    r = g.draw(3 * n + 8)
    for i in range(n):
        if r[3 * i] % 100 < int(sub_rate * 100):
            t[i] = AA20[int(r[3 * i + 1]) % 20]
    if n > 4 * indel:
        a = int(r[3 * n]) % (n // 2 - indel) + 1
        b = n // 2 + int(r[3 * n + 1]) % (n // 2 - indel - 1)
        ins = [AA20[int(r[3 * n + 2 + k]) % 20] for k in range(indel)]
        t = t[:a] + t[a + indel:b] + ins + t[b:]
    return q, "".join(t)[:n].ljust(n, "A")


def random_profile(seed, n):
    """Synthetic HMAP-style profile of n residues + 2 sentinels: per position a 20-vector of U*U
    normalised to sum 1, an SSE triple normalised to sum 1, a confidence U(0,1).  float32 throughout."""
    g = MT19937(seed)
    L = n + 2
    u = g.draw(L * 40 + L * 3 + L).astype(np.float64) / 4294967296.0
    a = (u[:L * 20] * u[L * 20:L * 40]).reshape(L, 20) + 1e-3
    aa = (a / a.sum(axis=1, keepdims=True)).astype(np.float32)
    s = u[L * 40:L * 43].reshape(L, 3) + 0.05
    sse = (s / s.sum(axis=1, keepdims=True)).astype(np.float32)
    conf = u[L * 43:L * 44].astype(np.float32)
    return {"aa": aa, "sse": sse, "conf": conf}


def make_subopt_regions(T, regs):
    """SuboptFlags for config 4 as the drivers make them (gn2.cpp:268-283): T template positions divided into `regs`
    alternating regions; the float comparisons are the reference's (fp32)."""
    length = np.float32(T) / np.float32(regs)
    flags = np.zeros(T, dtype=np.uint8)
    flag = True
    place = np.float32(length)
    for i in range(T):
        flags[i] = flag
        if np.float32(i) > place:
            flag = not flag
            place = np.float32(place + length)
    flags[T - 1] = 1
    return flags
