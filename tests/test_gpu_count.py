"""GPU parity: k-mer counting (bin/KPopCount.ml:36-50) through the C ABI vs the oracle.  Bit-exact."""
import numpy as np
import pytest

from conftest import GOLDEN, concat, load_golden

pytestmark = pytest.mark.gpu


def spectra_equal(a, b):
    (ha, ca, oa), (hb, cb, ob) = a, b
    assert oa.tolist() == ob.tolist()
    assert np.array_equal(ha, hb)
    assert np.array_equal(ca, cb)


def random_reads(rng, n, lo, hi, p_n=0.01):
    seqs = []
    for _ in range(n):
        L = int(rng.randint(lo, hi + 1))
        s = rng.choice(list("ACGTNacgtRY-"), size=L, p=[(1 - p_n * 4) / 4] * 4 + [p_n] + [p_n / 2] * 4 + [p_n / 3] * 3)
        seqs.append("".join(s))
    return seqs


def test_count_golden_vectors(kpop, oracle):
    g = load_golden("count_small.json")
    seqs = [s for _, s in g["reads"]]
    bases, offs = concat(seqs)
    for case in g["cases"]:
        k = case["k"]
        content = kpop.DNA_DS if case["content"] == "DNA-ds" else kpop.DNA_SS
        h, c, o = kpop.count_reads(bases, offs, k, content)
        for r in range(len(seqs)):
            got = [[oracle.to_hex(a, k), int(b)] for a, b in zip(h[int(o[r]):int(o[r + 1])], c[int(o[r]):int(o[r + 1])])]
            assert got == case["spectra"][r], (g["reads"][r][0], k, case["content"])


@pytest.mark.parametrize("k", [1, 4, 10, 12, 15, 16, 17, 24, 30])
@pytest.mark.parametrize("content", [0, 1])
def test_count_random_ragged(kpop, oracle, k, content):
    rng = np.random.RandomState(100 + k)
    seqs = ["", "A", "ACGT" * 3] + random_reads(rng, 300, 0, 180)
    bases, offs = concat(seqs)
    spectra_equal(kpop.count_reads(bases, offs, k, content), oracle.count_reads(bases, offs, k, content))


@pytest.mark.parametrize("max_len", [60, 75, 139, 267, 523])
def test_count_every_keys_per_lane_variant(kpop, oracle, max_len):
    """Read lengths that select R = 1, 2, 4, 8 keys per lane, up to the last window that fits."""
    rng = np.random.RandomState(max_len)
    k = 12
    seqs = random_reads(rng, 64, max(0, max_len - 40), max_len, p_n=0.004) + ["ACGT" * (max_len // 4)]
    seqs.append("".join(rng.choice(list("ACGT"), size=max_len)))
    bases, offs = concat(seqs)
    spectra_equal(kpop.count_reads(bases, offs, k), oracle.count_reads(bases, offs, k))


def test_count_headline_shape_10k_k10(kpop, oracle):
    """BASELINE config 2: synthetic 10k x 150 bp, k=10, bit-exact k-mer check."""
    bases, offs = oracle.synth_reads(0x4B506F70, 10000, 150)
    spectra_equal(kpop.count_reads(bases, offs, 10), oracle.count_reads(bases, offs, 10))


def test_count_offsets_not_starting_at_zero(kpop, oracle):
    bases, offs = oracle.synth_reads(5, 50, 100)
    sub = offs[10:31].copy()
    h, c, o = kpop.count_reads(bases, sub, 12)
    ho, co, oo = oracle.count_reads(bases, sub, 12)
    spectra_equal((h, c, o), (ho, co, oo))


def test_count_errors(kpop):
    bases, offs = concat(["ACGTACGTACGT"])
    for k in (0, 31):
        with pytest.raises(kpop.KPopError):
            kpop.count_reads(bases, offs, k)
    with pytest.raises(kpop.KPopError):
        kpop.count_reads(bases, offs, 13, content=kpop.PROTEIN)  # protein: k <= 12 (bin/KPopCount.ml:113)
    with pytest.raises(kpop.KPopError):
        kpop.count_reads(bases, offs, 3, content=3)
    with pytest.raises(kpop.KPopError) as e:
        kpop.count_reads(bases, offs, 3, capacity=1)
    assert e.value.code == -2  # KPOP_ERR_CAPACITY
    h, c, o = kpop.count_reads(np.zeros(0, np.uint8), np.zeros(1, np.uint64), 5)
    assert len(h) == 0 and o.tolist() == [0]


@pytest.fixture(params=[(1, 1), (1, 0), (1, 2), (1, 3), (0, 1)], ids=["histogram-lds", "histogram-direct", "histogram-combined-chunks", "histogram-partitioned", "sort"])
def merged_path(request, kpop):
    """-l by histogram (hashes of up to 26 bits; the default there) -- staged through LDS as the batch suggests (private tables up
    to k = 7, combined chunks of assemblies of one organism, partition-then-count for what does not repeat; the default), with
    direct global atomics, with every chunk combined, always partitioned -- and by device-wide sort (everything else)"""
    from kpop_amd import api
    api.tune("hist", request.param[0])
    api.tune("histlds", request.param[1])
    yield request.param
    api.tune("hist", 1)
    api.tune("histlds", 1)


def test_count_merged_golden_vectors(kpop, oracle, merged_path):
    """-l: one spectrum for all reads (bin/KPopCount.ml:60)."""
    g = load_golden("count_small.json")
    seqs = [s for _, s in g["reads"]]
    bases, offs = concat(seqs)
    for case in g["cases"]:
        k = case["k"]
        content = kpop.DNA_DS if case["content"] == "DNA-ds" else kpop.DNA_SS
        h, c, o = kpop.count_reads(bases, offs, k, content, per_read=False)
        assert o.tolist() == [0, len(case["merged"])]
        assert [[oracle.to_hex(a, k), int(b)] for a, b in zip(h, c)] == case["merged"], (k, case["content"])


@pytest.mark.parametrize("k", [5, 12, 13, 16, 24])
def test_count_merged_100k_reads(kpop, oracle, k, merged_path):
    bases, offs = oracle.synth_reads(0x4B506F70, 100000, 150)
    spectra_equal(kpop.count_reads(bases, offs, k, per_read=False), oracle.count_reads(bases, offs, k, per_read=False))


def test_count_merged_genomes_and_ragged(kpop, oracle, merged_path):
    """a few long sequences among short and empty ones, either strand convention, through both merged paths"""
    rng = np.random.RandomState(3)
    seqs = ["".join(rng.choice(list("ACGTN"), size=n, p=[0.249, 0.249, 0.249, 0.249, 0.004])) for n in (30000, 0, 5, 700, 100000, 12, 4097, 4108)]
    bases, offs = concat(seqs)
    for k in (7, 12, 13):
        for content in (kpop.DNA_DS, kpop.DNA_SS):
            spectra_equal(kpop.count_reads(bases, offs, k, content, per_read=False), oracle.count_reads(bases, offs, k, content, per_read=False))


def test_count_merged_assemblies_of_one_organism(kpop, oracle, merged_path):
    """-l over 70 mutated copies of one 9 kb sequence and 40 unrelated ones (BASELINE config 3's kind of batch): the chunks
    of the LDS-staged histogram see the same k-mers across sequences (and, for the unrelated ones, nothing twice)"""
    rng = np.random.RandomState(11)
    ref = rng.choice(list("ACGT"), size=9000)
    seqs = []
    for i in range(70):
        m = ref.copy()
        hit = rng.rand(len(m)) < 0.002
        m[hit] = rng.choice(list("ACGTN"), size=int(hit.sum()))
        seqs.append("".join(m[: len(m) - int(rng.randint(0, 50))]))
    seqs += ["".join(rng.choice(list("ACGT"), size=int(rng.randint(4200, 12000)))) for _ in range(40)]
    bases, offs = concat(seqs)
    for k in (4, 7, 8, 12):
        spectra_equal(kpop.count_reads(bases, offs, k, per_read=False), oracle.count_reads(bases, offs, k, per_read=False))
    pb, po = concat(["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWYX"), size=5000)) for _ in range(40)])
    for k in (2, 4):
        spectra_equal(kpop.count_reads(pb, po, k, kpop.PROTEIN, per_read=False), oracle.count_reads(pb, po, k, oracle.PROTEIN, per_read=False))


@pytest.mark.parametrize("k", [8, 12, 21, 30])
def test_count_genomes_per_read(kpop, oracle, k):
    """-L on sequences beyond one wavefront's 512 windows: wuhan (29,903 bp), a 100 kb sequence with Ns,
    short reads in the same batch; k=30 forces sub-batches of 8 spectra (63-bit composite keys)."""
    rng = np.random.RandomState(k)
    wuhan = "".join(l.strip() for l in open(GOLDEN + "/wuhan.fasta") if not l.startswith(">"))
    big = "".join(rng.choice(list("ACGT"), size=100000))
    big = big[:50000] + "NNNNN" + big[50005:]
    seqs = [wuhan, "ACGT" * 30, big, "", "ACG", wuhan[:600]] + random_reads(rng, 14, 0, 700)
    bases, offs = concat(seqs)
    spectra_equal(kpop.count_reads(bases, offs, k), oracle.count_reads(bases, offs, k))
    spectra_equal(kpop.count_reads(bases, offs, k, kpop.DNA_SS), oracle.count_reads(bases, offs, k, oracle.DNA_SS))


def test_count_wuhan_golden(kpop, pyref):
    import hashlib
    g = load_golden("wuhan_counts.json")
    seq = "".join(l.strip() for l in open(GOLDEN + "/wuhan.fasta") if not l.startswith(">"))
    bases, offs = concat([seq])
    for case in g["cases"]:
        k = case["k"]
        h, c, o = kpop.count_reads(bases, offs, k)
        assert len(h) == case["n_distinct"] and int(c.sum()) == case["total"]
        text = pyref.spectrum_text("MN908947.3", {int(a): int(b) for a, b in zip(h, c)}, k)
        assert hashlib.sha256(text.encode()).hexdigest() == case["spectrum_text_sha256"]


def test_count_protein_golden_and_random(kpop, oracle):
    """-C protein (KMers.ProteinHash under the declared 5-bit encoding): per-sequence and merged spectra against the
    golden fixture, then random proteomes -- short sequences (one wavefront each), long ones (sort path), 32- and
    64-bit keys (k <= 6 / k > 6)."""
    g = load_golden("count_protein.json")
    seqs = [s for _, s in g["reads"]]
    bases, offs = concat(seqs)
    for case in g["cases"]:
        k = case["k"]
        h, c, o = kpop.count_reads(bases, offs, k, kpop.PROTEIN)
        for r in range(len(seqs)):
            got = [[oracle.to_hex(a, k, oracle.PROTEIN), int(b)] for a, b in zip(h[int(o[r]):int(o[r + 1])], c[int(o[r]):int(o[r + 1])])]
            assert got == case["spectra"][r], (k, g["reads"][r][0])
        hm, cm, om = kpop.count_reads(bases, offs, k, kpop.PROTEIN, per_read=False)
        assert [[oracle.to_hex(a, k, oracle.PROTEIN), int(b)] for a, b in zip(hm, cm)] == case["merged"], k
    rng = np.random.RandomState(99)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYXBZ*acdefghiklmnpqrstvwy-", dtype=np.uint8)
    lens = [int(x) for x in rng.randint(0, 700, size=300)] + [5000, 40000]
    prot = [bytes(aa[rng.randint(0, len(aa), size=n)]).decode() for n in lens]
    pb, po = concat(prot)
    for k in (2, 6, 7, 12):
        spectra_equal(kpop.count_reads(pb, po, k, kpop.PROTEIN), oracle.count_reads(pb, po, k, oracle.PROTEIN))
        spectra_equal(kpop.count_reads(pb, po, k, kpop.PROTEIN, per_read=False), oracle.count_reads(pb, po, k, oracle.PROTEIN, per_read=False))


@pytest.mark.parametrize("k", [4, 12, 15])
def test_count_assemblies_block_sort_equals_device_sort_and_oracle(kpop, oracle, k):
    """-L on sequences of 513..32,768 windows: one block per sequence sorting in LDS (the default there) against the
    device-wide sort and the oracle; lengths around the powers of two, Ns, a homopolymer (one run of 30,000), protein too"""
    from kpop_amd import api
    rng = np.random.RandomState(k + 40)
    wuhan = "".join(l.strip() for l in open(GOLDEN + "/wuhan.fasta") if not l.startswith(">"))
    lens = [513 + k, 1023 + k, 1024 + k, 1025 + k, 5000, 16383 + k, 16384 + k, 32768 + k - 1, 700, 0, 3]
    seqs = [wuhan, "A" * 30000] + ["".join(rng.choice(list("ACGTN"), size=n, p=[0.248, 0.248, 0.248, 0.248, 0.008])) for n in lens]
    bases, offs = concat(seqs)
    for content in (kpop.DNA_DS, kpop.DNA_SS):
        want = oracle.count_reads(bases, offs, k, content)
        res = {}
        for flag in (1, 0):
            api.tune("blocksort", flag)
            res[flag] = kpop.count_reads(bases, offs, k, content)
        api.tune("blocksort", 1)
        spectra_equal(res[1], want)
        spectra_equal(res[0], want)
    if k <= 6:
        prot = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWYX"), size=n)) for n in (600, 3000, 20000)]
        pb, po = concat(prot)
        spectra_equal(kpop.count_reads(pb, po, k, kpop.PROTEIN), oracle.count_reads(pb, po, k, oracle.PROTEIN))


@pytest.mark.parametrize("guess", [1, 0])
def test_count_merged_when_the_sample_does_not_speak_for_the_batch(kpop, oracle, guess):
    """-l on reads that do not repeat, partition path: the buckets' rooms come from ONE item in 32 (kpop_tune("histguess", 1), the
    default).  Here every 32nd read -- the sampled ones -- is random and the 31 between them are copies of a few low-complexity
    reads: the buckets those fall into overflow their guessed rooms, the partition pass says so and the call goes back to the
    exact count.  The same spectrum either way, and as the oracle's; then an ordinary read set, where the guess holds."""
    from kpop_amd import api
    rng = np.random.RandomState(5)
    n = 40000
    rnd = ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(n // 32 + 1)]
    few = ["ACGTTGCA" * 19, "A" * 150, ("AC" * 75), "".join(rng.choice(list("ACGT"), size=150))]
    seqs = [rnd[i // 32] if i % 32 == 0 else few[i % 4] for i in range(n)]
    bases, offs = concat(seqs)
    api.tune("histlds", 3)  # (always the partition path: repeats would otherwise send the batch elsewhere)
    api.tune("histguess", guess)
    try:
        for k in (12, 13):
            spectra_equal(kpop.count_reads(bases, offs, k, per_read=False), oracle.count_reads(bases, offs, k, per_read=False))
        b2, o2 = oracle.synth_reads(0x77, 60000, 150)
        spectra_equal(kpop.count_reads(b2, o2, 12, per_read=False), oracle.count_reads(b2, o2, 12, per_read=False))
    finally:
        api.tune("histlds", 1)
        api.tune("histguess", 1)
