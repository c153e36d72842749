"""Seeded synthetic inputs for the BASELINE.json configs.

There is no network for real FASTQ / enwik data, so bench.py and the tests use
these generators.  They are deterministic functions of (seed, size).

* ``fastq_like``  -- config 2/3/4: 4-line records, 150-bp reads sampled from a
  seeded random ACGT "genome" with 1 % substitutions, header
  ``@SRR000001.<i> <i>/1``, 150 quality symbols from a 26-character skewed
  alphabet.  libdeflate 1.23 level 1 compresses it to 0.274, level 6 to 0.259
  (0xff00-byte blocks; measured in the build container with the reference's own
  libdeflate 1.23).
* ``text_like``   -- config 5: Zipf-distributed words over a 50k vocabulary with
  wiki-ish markup (zlib-6 ratio ~0.33-0.36).
* ``random_bytes`` -- config 1 stand-in for /dev/urandom (incompressible).
"""
import numpy as np

QUAL_REPEAT = 0.6
QUAL_SKEW = 2.0


def fastq_like(nbytes, seed=1234, genome_len=5_000_000, first_record=1):
    rng = np.random.default_rng(seed)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, genome_len)]
    read_len = 150
    # one record is 2*150 + 4 + header (17..40 bytes): never fewer than 321 bytes
    nrec = nbytes // 321 + 2
    pos = rng.integers(0, genome_len - read_len, nrec)
    reads = genome[pos[:, None] + np.arange(read_len)[None, :]]
    sub = rng.random((nrec, read_len)) < 0.01
    reads = np.where(sub, np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (nrec, read_len))], reads)
    qual_alphabet = np.frombuffer(b"F:,#IHGEDCBA@?>=<;98765432", dtype=np.uint8)
    w = np.array([2.0 ** (-QUAL_SKEW * k) for k in range(26)])
    w /= w.sum()
    qsym = rng.choice(26, size=(nrec, read_len), p=w)
    # Illumina-like persistence: a quality value repeats its left neighbour
    # with probability QUAL_REPEAT (forward fill along the read)
    keep = rng.random((nrec, read_len)) >= QUAL_REPEAT
    keep[:, 0] = True
    src_col = np.maximum.accumulate(np.where(keep, np.arange(read_len)[None, :], 0), axis=1)
    qual = qual_alphabet[np.take_along_axis(qsym, src_col, axis=1)]
    parts = []
    total = 0
    nl = b"\n"
    for i in range(nrec):
        r = first_record + i
        rec = b"@SRR000001.%d %d/1\n" % (r, r) + reads[i].tobytes() + b"\n+\n" + qual[i].tobytes() + nl
        parts.append(rec)
        total += len(rec)
        if total >= nbytes:
            break
    blob = b"".join(parts)
    assert len(blob) >= nbytes
    return np.frombuffer(blob[:nbytes], dtype=np.uint8).copy()


def text_like(nbytes, seed=4321, vocab=50_000):
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    lw = np.array([1.0 / (k + 1) ** 0.6 for k in range(26)])
    lw /= lw.sum()
    lens = rng.integers(2, 12, vocab)
    words = [letters[rng.choice(26, size=int(n), p=lw)].tobytes() for n in lens]
    markup = [b"[[", b"]]", b"'''", b"&quot;", b"==", b"{{", b"}}", b"\n", b"\n\n", b"<ref>", b"</ref>", b"|"]
    nwords = nbytes // 5 + 16
    zw = 1.0 / (np.arange(vocab) + 1.5) ** 1.1
    zw /= zw.sum()
    ids = rng.choice(vocab, size=nwords, p=zw)
    mk = rng.random(nwords)
    mki = rng.integers(0, len(markup), nwords)
    out = []
    total = 0
    for k in range(nwords):
        wbytes = words[ids[k]]
        if mk[k] < 0.08:
            wbytes = markup[mki[k]] + wbytes
        out.append(wbytes)
        total += len(wbytes) + 1
        if total >= nbytes + 16:
            break
    blob = b" ".join(out)
    while len(blob) < nbytes:              # Zipf tail shorter than planned: pad with more of the same
        blob += b" " + blob[: nbytes - len(blob)]
    return np.frombuffer(blob[:nbytes], dtype=np.uint8).copy()


def random_bytes(nbytes, seed=99):
    return np.random.default_rng(seed).integers(0, 256, nbytes, dtype=np.uint8)
