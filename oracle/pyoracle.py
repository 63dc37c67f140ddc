"""oracle/pyoracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Second, independent restatement (pure Python, small inputs only) of the MinHash hot
path of luizirber/sourmash-rust.  It exists to cross-check the C oracle
(oracle/sourmash_oracle.c) and to generate the committed golden vectors
(tests/golden/make_golden.py).  Only tests/ may import it.

Written from the reference's behaviour, not from the C oracle: it uses Python ints,
bytes.translate and bisect, so a slip shared with the C file is unlikely.

Reference lines followed (relative to the reference root):
  _hash_murmur            src/lib.rs:33-35  (+ crate murmurhash3 ~0.0.5, Cargo.toml:49)
  check_compatible        src/lib.rs:176-190
  add_hash                src/lib.rs:192-245
  add_sequence            src/lib.rs:252-305
  merge                   src/lib.rs:307-403
  count_common            src/lib.rs:428-436
  intersection_size       src/lib.rs:470-499
  compare                 src/lib.rs:501-508
  revcomp / to_aa / _checkdna   src/lib.rs:677-689, 691-793, 795-804
  containment             src/index.rs:146-154
"""
from bisect import bisect_left

M64 = (1 << 64) - 1
C1 = 0x87C37B91114253D5
C2 = 0x4CF5AD432745937F


class OraclePanic(Exception):
    """Where the Rust reference would panic."""


class OracleError(Exception):
    def __init__(self, code, message=""):
        super().__init__(message)
        self.code = code
        self.message = message


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _fmix(k):
    k ^= k >> 33
    k = (k * 0xFF51AFD7ED558CCD) & M64
    k ^= k >> 33
    k = (k * 0xC4CEB9FE1A85EC53) & M64
    k ^= k >> 33
    return k


def murmur3_x64_128(data, seed):
    data = bytes(data)
    n = len(data)
    h1 = h2 = seed & M64
    full = n - (n % 16)
    for off in range(0, full, 16):
        k1 = int.from_bytes(data[off:off + 8], "little")
        k2 = int.from_bytes(data[off + 8:off + 16], "little")
        k1 = (_rotl((k1 * C1) & M64, 31) * C2) & M64
        h1 ^= k1
        h1 = (_rotl(h1, 27) + h2) & M64
        h1 = (h1 * 5 + 0x52DCE729) & M64
        k2 = (_rotl((k2 * C2) & M64, 33) * C1) & M64
        h2 ^= k2
        h2 = (_rotl(h2, 31) + h1) & M64
        h2 = (h2 * 5 + 0x38495AB5) & M64
    tail = data[full:]
    if len(tail) > 8:
        k2 = int.from_bytes(tail[8:], "little")
        h2 ^= (_rotl((k2 * C2) & M64, 33) * C1) & M64
    if len(tail) > 0:
        k1 = int.from_bytes(tail[:8], "little")
        h1 ^= (_rotl((k1 * C1) & M64, 31) * C2) & M64
    h1 ^= n
    h2 ^= n
    h1 = (h1 + h2) & M64
    h2 = (h2 + h1) & M64
    h1 = _fmix(h1)
    h2 = _fmix(h2)
    h1 = (h1 + h2) & M64
    h2 = (h2 + h1) & M64
    return h1, h2


def hash_murmur(kmer, seed=42):
    return murmur3_x64_128(kmer, seed)[0]


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")
_UPPER = bytes.maketrans(bytes(range(ord("a"), ord("z") + 1)), bytes(range(ord("A"), ord("Z") + 1)))
_DNA = frozenset(b"ACGTacgt")


def revcomp(seq):
    return bytes(seq).translate(_COMP)[::-1]


def _build_codon_table():
    bases = "TCAG"
    aas = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
    table = {}
    i = 0
    for a in bases:
        for b in bases:
            for c in bases:
                table[(a + b + c).encode()] = ord(aas[i])
                i += 1
    return table


CODONS = _build_codon_table()


def to_aa(seq):
    out = bytearray()
    for i in range(0, len(seq) - 2, 3):
        chunk = bytes(seq[i:i + 3])
        try:
            chunk.decode("utf-8")
        except UnicodeDecodeError:
            raise OraclePanic("from_utf8 on codon")
        aa = CODONS.get(chunk)
        if aa is not None:
            out.append(aa)
    return bytes(out)


class MinHash:
    def __init__(self, num, ksize, is_protein=False, seed=42, max_hash=0, track_abundance=False):
        self.num = num
        self.ksize = ksize
        self.is_protein = bool(is_protein)
        self.seed = seed
        self.max_hash = max_hash
        self.mins = []
        self.abunds = [] if track_abundance else None

    def copy(self):
        o = MinHash(self.num, self.ksize, self.is_protein, self.seed, self.max_hash, False)
        o.mins = list(self.mins)
        o.abunds = None if self.abunds is None else list(self.abunds)
        return o

    def check_compatible(self, other):
        if self.ksize != other.ksize:
            raise OracleError(101, "different ksizes cannot be compared")
        if self.is_protein != other.is_protein:
            raise OracleError(102, "DNA/prot minhashes cannot be compared")
        if self.max_hash != other.max_hash:
            raise OracleError(103, "mismatch in max_hash; comparison fail")
        if self.seed != other.seed:
            raise OracleError(104, "mismatch in seed; comparison fail")

    def add_hash(self, h):
        if self.max_hash != 0 and h > self.max_hash:
            return
        if not self.mins:
            self.mins.append(h)
            if self.abunds is not None:
                self.abunds.append(1)
            return
        if h <= self.max_hash or self.mins[-1] > h or len(self.mins) < self.num:
            pos = bisect_left(self.mins, h)
            if pos == len(self.mins):
                self.mins.append(h)
                if self.abunds is not None:
                    self.abunds.append(1)
            elif self.mins[pos] != h:
                self.mins.insert(pos, h)
                if self.abunds is not None:
                    if pos > len(self.abunds):
                        raise OraclePanic("Vec::insert out of range")
                    self.abunds.insert(pos, 1)
                if self.num != 0 and len(self.mins) > self.num:
                    self.mins.pop()
                    if self.abunds:
                        self.abunds.pop()
            elif self.abunds is not None:
                if pos >= len(self.abunds):
                    raise OraclePanic("index out of range")
                self.abunds[pos] += 1

    def add_word(self, word):
        self.add_hash(hash_murmur(word, self.seed))

    def add_many(self, hashes):
        for h in hashes:
            self.add_hash(h)

    def add_many_with_abund(self, items):
        # src/lib.rs:419-426
        for h, n in items:
            for _ in range(n):
                self.add_hash(h)

    def add_sequence(self, seq, force=False):
        s = bytes(seq).translate(_UPPER)
        k = self.ksize
        if len(s) < k:
            return
        if not self.is_protein:
            if k == 0:
                raise OraclePanic("windows(0)")
            for i in range(len(s) - k + 1):
                kmer = s[i:i + k]
                if all(c in _DNA for c in kmer):
                    rc = revcomp(kmer)
                    self.add_word(kmer if kmer < rc else rc)
                elif not force:
                    try:
                        msg = kmer.decode("utf-8")
                    except UnicodeDecodeError:
                        raise OraclePanic("from_utf8 on k-mer")
                    raise OracleError(1101, "invalid DNA character in input k-mer: " + msg)
        else:
            rc = revcomp(s)
            aak = k // 3
            for i in range(3):
                for strand in (s, rc):
                    aa = to_aa(strand[i:])
                    if aak == 0:
                        raise OraclePanic("windows(0)")
                    for w in range(len(aa) - aak + 1):
                        self.add_word(aa[w:w + aak])

    def merge(self, other):
        self.check_compatible(other)
        a, b = self.mins, other.mins
        sa = None if self.abunds is None else iter(self.abunds)
        oa = None if other.abunds is None else iter(other.abunds)
        merged, mab = [], []
        i = j = 0
        while i < len(a):
            if j >= len(b):
                merged.extend(a[i:])
                i = len(a)
                if sa is not None:
                    mab.extend(sa)
                    sa = iter(())
                break
            if b[j] < a[i]:
                merged.append(b[j])
                j += 1
                if oa is not None:
                    v = next(oa, None)
                    if v is not None:
                        mab.append(v)
            elif b[j] == a[i]:
                merged.append(b[j])
                i += 1
                j += 1
                if oa is not None:
                    v = next(oa, None)
                    if v is not None and sa is not None:
                        s_ = next(sa, None)
                        if s_ is not None:
                            mab.append(v + s_)
            else:
                merged.append(a[i])
                i += 1
                if sa is not None:
                    v = next(sa, None)
                    if v is not None:
                        mab.append(v)
        merged.extend(b[j:])
        if oa is not None:
            mab.extend(oa)
        if len(merged) < self.num or self.num == 0:
            self.mins = merged
        else:
            self.mins = merged[:self.num]
        self.abunds = mab

    def count_common(self, other):
        self.check_compatible(other)
        return _two_pointer(self.mins, other.mins)[0]

    def intersection_size(self, other):
        self.check_compatible(other)
        comb = MinHash(self.num, self.ksize, self.is_protein, self.seed, self.max_hash,
                       self.abunds is not None)
        comb.merge(self)
        comb.merge(other)
        i1 = _two_pointer(self.mins, other.mins)[1]
        return _two_pointer(i1, comb.mins)[0], len(comb.mins)

    def compare(self, other):
        self.check_compatible(other)
        common, size = self.intersection_size(other)
        return common / max(1, size)

    def containment(self, other):
        c = self.count_common(other)
        return c / len(self.mins) if self.mins else float("nan")


def _two_pointer(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        if a[i] < b[j]:
            i += 1
        elif a[i] > b[j]:
            j += 1
        else:
            out.append(a[i])
            i += 1
            j += 1
    return len(out), out


def splitmix64(seed, index):
    z = (seed + (index + 1) * 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def synth_dna(start, length, seed, n_every=0):
    out = bytearray(length)
    for i in range(length):
        p = start + i
        w = splitmix64(seed, p >> 5)
        c = b"ACGT"[(w >> (2 * (p & 31))) & 3]
        if n_every and p % n_every == n_every - 1:
            c = ord("N")
        out[i] = c
    return bytes(out)
