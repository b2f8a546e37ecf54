"""Deterministic synthetic read sets.

Counter-based (splitmix64) so that the numpy version here, the C version kept with the CPU
baseline and the HIP kernel ``mcom_synth_reads`` (minicom_amd/csrc/reads.hip) produce byte-identical reads for the same
(seed, n_reads, read_len, coverage) without sharing state:

  genome length G = max(n*L/coverage, L+1); base g        = key(0, g) & 3
  read r:  start = key(1, r) % (G - L + 1);  strand       = key(2, r) & 1
           base i substituted when (key(3, r*L+i) & 0xFFFFFF) < sub_rate*2^24,
           new base = (old + 1 + ((key >> 24) % 3)) & 3
           strand 1 -> reverse complement
  "plumbing" extras (only when plumbing=True), c = key(4, r) % 10000:
           c < 100  -> 1..3 bases replaced by 'N';  100 <= c < 105 -> poly-A with <=3 other bases
           105 <= c < 110 -> poly-T likewise;  c == 110 -> all 'N';  c == 111 -> all 'A'; c == 112 -> all 'T'
           113 <= c < 118 -> more than 0.4*L bases replaced by 'N'
  key(s, i) = sm64(sm64(seed + s) + i)
"""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def sm64(z: np.ndarray) -> np.ndarray:
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def key(seed: int, stream: int, idx) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = sm64(np.uint64((seed + stream) & 0xFFFFFFFFFFFFFFFF))
        return sm64(base + np.asarray(idx, dtype=np.uint64))


def genome_len(n_reads: int, read_len: int, coverage: int = 30) -> int:
    return max(n_reads * read_len // coverage, read_len + 1)


ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def synth_reads(seed: int, n_reads: int, read_len: int, coverage: int = 30,
                sub_rate: float = 0.005, plumbing: bool = False,
                first: int = 0, count: int | None = None) -> np.ndarray:
    """Return reads [count, read_len] as uint8 ASCII (rows ``first .. first+count`` of the set)."""
    L = read_len
    if count is None:
        count = n_reads - first
    G = genome_len(n_reads, L, coverage)
    r = np.arange(first, first + count, dtype=np.uint64)
    start = key(seed, 1, r) % np.uint64(G - L + 1)
    strand = (key(seed, 2, r) & np.uint64(1)).astype(bool)
    pos = start[:, None] + np.arange(L, dtype=np.uint64)[None, :]
    base = (key(seed, 0, pos) & np.uint64(3)).astype(np.uint8)
    u = key(seed, 3, r[:, None] * np.uint64(L) + np.arange(L, dtype=np.uint64)[None, :])
    thr = np.uint64(int(sub_rate * (1 << 24)))
    sub = (u & np.uint64(0xFFFFFF)) < thr
    nb = (base + np.uint8(1) + ((u >> np.uint64(24)) % np.uint64(3)).astype(np.uint8)) & np.uint8(3)
    base = np.where(sub, nb, base)
    rc = (np.uint8(3) - base)[:, ::-1]
    base = np.where(strand[:, None], rc, base)
    out = ACGT[base]
    if plumbing:
        c = (key(seed, 4, r) % np.uint64(10000)).astype(np.int64)
        for j in np.nonzero(c < 118)[0]:
            cj = int(c[j])
            rr = int(r[j])
            h = [int(x) for x in key(seed, 5, np.arange(rr * 8, rr * 8 + 8, dtype=np.uint64))]
            if cj < 100:
                for t in range(1 + h[0] % 3):
                    out[j, h[1 + t] % L] = ord("N")
            elif cj < 110:
                ch = ord("A") if cj < 105 else ord("T")
                keep = [(h[1 + t] % L, out[j, h[1 + t] % L]) for t in range(h[0] % 4)]
                out[j, :] = ch
                for p, b in keep:
                    out[j, p] = b
            elif cj == 110:
                out[j, :] = ord("N")
            elif cj == 111:
                out[j, :] = ord("A")
            elif cj == 112:
                out[j, :] = ord("T")
            else:
                nN = int(0.4 * L) + 1 + h[0] % 5
                idx = key(seed, 6, np.arange(rr * 512, rr * 512 + nN, dtype=np.uint64)) % np.uint64(L)
                out[j, idx.astype(np.int64)] = ord("N")
    return np.ascontiguousarray(out)


def repeat_rich_reads(n: int, L: int, sub_rate: float = 0.004, seed: int = 97, genome: int = 200_000) -> np.ndarray:
    """A repeat-rich read set: a short genome with a 2 kb segment in forty (partly mutated) copies, a tandem repeat of a
    37-base unit, a poly-A and an (AT)n stretch, sampled at very high coverage -- groups of tens of thousands of reads, long
    runs of equal minimizers in the contig index, contigs of 10^5 members, several Stage-2 passes at higher error rates."""
    rng = np.random.default_rng(seed)
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    G = genome
    g = ACGT[rng.integers(0, 4, G)]
    rep = ACGT[rng.integers(0, 4, 2000)]
    for at in rng.integers(0, G - 2000, 40):
        c = rep.copy()
        for q in rng.integers(0, 2000, int(rng.integers(0, 6))):
            c[q] = ACGT[rng.integers(0, 4)]
        g[at:at + 2000] = c
    unit = ACGT[rng.integers(0, 4, 37)]
    g[50_000:50_000 + 37 * 60] = np.tile(unit, 60)
    g[120_000:120_400] = ord("A"); g[130_000:130_300] = np.tile(np.frombuffer(b"AT", dtype=np.uint8), 150)
    start = rng.integers(0, G - L + 1, n)
    reads = g[start[:, None] + np.arange(L)[None, :]]
    sub = rng.random((n, L)) < sub_rate
    reads = np.where(sub, ACGT[rng.integers(0, 4, (n, L))], reads)
    rc = rng.random(n) < 0.5
    reads[rc] = comp[reads[rc]][:, ::-1]
    return np.ascontiguousarray(reads)


def write_fastq_fast(path: str, reads: np.ndarray, tricky_quality: bool = True) -> None:
    """FASTQ file of four-line records, names @r0, @r1, ... (records are NOT of one size), written with numpy in blocks of equal
    name width: millions of reads per second.  tricky_quality: every third quality line starts with '@', every fifth with '+' (the
    characters a parser that looks for record starts must not trust)."""
    n, L = reads.shape
    with open(path, "wb") as f:
        lo = 0
        while lo < n:
            digits = len(str(lo))
            hi = min(n, 10 ** digits)
            for a in range(lo, hi, 1 << 20):
                b = min(hi, a + (1 << 20))
                m = b - a
                rec = np.empty((m, 2 + digits + 1 + L + 3 + L + 1), dtype=np.uint8)
                rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
                ids = np.arange(a, b, dtype=np.int64)
                for d in range(digits):
                    rec[:, 2 + digits - 1 - d] = ord("0") + (ids // 10 ** d) % 10
                o = 2 + digits
                rec[:, o] = 10
                rec[:, o + 1:o + 1 + L] = reads[a:b]
                rec[:, o + 1 + L] = 10; rec[:, o + 2 + L] = ord("+"); rec[:, o + 3 + L] = 10
                q = rec[:, o + 4 + L:o + 4 + 2 * L]
                q[:] = ord("I")
                if tricky_quality:
                    q[ids % 3 == 0, 0] = ord("@"); q[ids % 5 == 0, 0] = ord("+")
                rec[:, o + 4 + 2 * L] = 10
                f.write(rec.tobytes())
            lo = hi


def write_fastq(path: str, reads: np.ndarray) -> None:
    n, L = reads.shape
    qual = b"I" * L
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b"@r%d\n" % i)
            f.write(reads[i].tobytes())
            f.write(b"\n+\n")
            f.write(qual)
            f.write(b"\n")
