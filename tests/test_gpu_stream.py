"""The reads stream between the two drop-ins (kpop_amd/host/fast_seq.h): `KPopCount -L | KPopTwistDB -k /dev/stdin` with
the counting deferred to KPopTwistDB must write what the text-spectra pipeline writes, byte for byte; and
kpop_spectra_twist must equal kpop_count_reads followed by kpop_twist, bit for bit, for any length and any n_dims."""
import os
import subprocess

import numpy as np
import pytest

from conftest import concat
from test_cli import COUNT, TWISTDB
from test_gpu_cli import make_twister, write_fasta

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(TWISTDB), reason="host CLIs not built")]


def _reads(rng, n_short=40, genomes=(700, 5000, 20000)):
    seqs = ["".join(rng.choice(list("ACGT"), size=int(rng.randint(0, 200)))) for _ in range(n_short)]
    seqs += ["ACGTNNACGTACGTTTGACCANGGT", "", "ACG", "acgtacgtacgtagctagctagctagcatcgat"]
    seqs += ["".join(rng.choice(list("ACGTN"), size=g, p=[0.249, 0.249, 0.249, 0.249, 0.004])) for g in genomes]
    return seqs


@pytest.mark.parametrize("d", [5, 9, 32, 33, 64, 100])
def test_spectra_twist_equals_count_then_twist(kpop, oracle, d):
    rng = np.random.RandomState(d)
    for k, genomes in ((5, (700,)), (11, ()), (12, (5000, 20000)), (17, (3000,))):
        if k <= 12:
            cols = oracle.enumerate_kmers(k)
            cols = cols[rng.rand(len(cols)) < 0.7]
        else:
            cols = np.unique(rng.randint(0, 1 << 34, size=5000).astype(np.uint64))
        T = oracle.synth_twister(7, d, cols)
        for load_k in sorted({k, k + (k % 2)}):  # a loader that inferred k from the name width holds the even k
            tw = kpop.Twister.load(T, cols, load_k)
            for seqs in (_reads(rng, genomes=()), _reads(rng, genomes=genomes)):
                if k > 12:  # plant some k-mers the twister knows
                    seqs = seqs + ["".join("ACGT"[(int(h) >> (2 * (k - 1 - i))) & 3] for i in range(k)) * 3 for h in cols[:50]]
                bases, offs = concat(seqs)
                for content in (kpop.DNA_DS, kpop.DNA_SS):
                    h, c, o = kpop.count_reads(bases, offs, k, content=content)
                    for normalize in (True, False):
                        want = tw.twist(h, c.astype(np.float64), o, normalize=normalize)
                        got = tw.spectra_twist(bases, offs, k, content=content, normalize=normalize)
                        assert np.array_equal(got, want), (d, k, load_k, content, normalize)
            tw.free()


def _pipeline(tmp_path, fasta, k, prefix, fmt, extra_count=(), table=False, content=None):
    env = dict(os.environ, KPOP_PIPE_FORMAT=fmt)
    cmd = [COUNT, "-k", str(k), "-L", "-f", str(fasta)] + list(extra_count)
    if content:
        cmd += ["-C", content]
    p1 = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    p2 = subprocess.run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", "/dev/stdin", "-O" if table else "-o", "t", str(tmp_path / prefix)],
                        stdin=p1.stdout, capture_output=True, timeout=300, env=env)
    p1.stdout.close()
    e1 = p1.stderr.read().decode()
    p1.wait()
    return p1.returncode, p2.returncode, e1, p2.stderr.decode()


@pytest.mark.parametrize("k,d,tw_k", [(5, 6, 5), (12, 64, 12), (11, 9, 11), (11, 40, 12), (9, 7, 12)])
def test_reads_stream_equals_text_spectra(tmp_path, oracle, k, d, tw_k):
    """tw_k: the k the twister's columns were made with (its names are ceil(tw_k/2) digits wide)"""
    rng = np.random.RandomState(k * 100 + d)
    make_twister(tmp_path, oracle, tw_k, d, keep=0.6 if tw_k < 12 else 0.02)
    seqs = _reads(rng, genomes=(800, 6000))
    reads = [("s%d extra words" % i if i % 3 else '"q%d"' % i, s) for i, s in enumerate(seqs)]
    fa = tmp_path / "in.fa"
    write_fasta(fa, reads, width=61)
    for fmt in ("reads", "text", "auto"):
        rc1, rc2, e1, e2 = _pipeline(tmp_path, fa, k, "out_" + fmt, fmt, extra_count=["-v"] if fmt != "text" else [])
        assert rc1 == 0 and rc2 == 0, (fmt, e1, e2)
        if fmt == "reads" or fmt == "auto":  # auto: the reader of the pipe IS our KPopTwistDB
            assert "handing it the reads" in e1, (fmt, e1)
    a = (tmp_path / "out_reads.KPopTwisted").read_bytes()
    assert a == (tmp_path / "out_text.KPopTwisted").read_bytes() == (tmp_path / "out_auto.KPopTwisted").read_bytes()
    assert len(a) > 100
    # single-stranded counting goes the same way
    for fmt in ("reads", "text"):
        rc1, rc2, e1, e2 = _pipeline(tmp_path, fa, k, "ss_" + fmt, fmt, table=True, content="DNA-ss")
        assert rc1 == 0 and rc2 == 0, (fmt, e1, e2)
    assert (tmp_path / "ss_reads.KPopTwisted.txt").read_bytes() == (tmp_path / "ss_text.KPopTwisted.txt").read_bytes()


def test_reads_stream_errors_and_other_readers(tmp_path, oracle):
    k, d = 5, 6
    make_twister(tmp_path, oracle, k, d)
    fa = tmp_path / "dup.fa"
    write_fasta(fa, [("a", "ACGTACGTACGT"), ("b", "ACGTTTGACGT"), ("a", "ACGTTGCAAC")])
    msgs = []
    for fmt in ("reads", "text"):
        rc1, rc2, e1, e2 = _pipeline(tmp_path, fa, k, "dup_" + fmt, fmt)
        assert rc2 == 1 and 'Duplicate_label("a")' in e2, (fmt, e2)
        msgs.append(e2.strip().splitlines()[-1])
    assert msgs[0] == msgs[1]
    # a reader that is not KPopTwistDB gets text, whatever sits further down the pipe
    fa2 = tmp_path / "ok.fa"
    write_fasta(fa2, [("a", "ACGTACGTACGT"), ("b", "ACGTTTGACGT")])
    r = subprocess.run("%s -k %d -L -f %s -v | cat | %s -I T %s -k /dev/stdin -O t %s" %
                       (COUNT, k, fa2, TWISTDB, tmp_path / "Classes", tmp_path / "viacat"), shell=True, capture_output=True, text=True)
    assert r.returncode == 0 and "handing it the reads" not in r.stderr, r.stderr
    rc1, rc2, e1, e2 = _pipeline(tmp_path, fa2, k, "direct", "auto", table=True)
    assert rc1 == 0 and rc2 == 0
    assert (tmp_path / "viacat.KPopTwisted.txt").read_bytes() == (tmp_path / "direct.KPopTwisted.txt").read_bytes()
    # two producers on one pipe: two streams back to back, then text from a third
    sp = tmp_path / "c.KPopSpectra.txt"
    fa3 = tmp_path / "c.fa"
    write_fasta(fa3, [("c", "TTGACCAGTACCA")])
    assert subprocess.run([COUNT, "-k", str(k), "-L", "-f", str(fa3), "-o", str(tmp_path / "c")]).returncode == 0
    r = subprocess.run("( %s -k %d -L -f %s; %s -k %d -L -f %s; cat %s ) | %s -I T %s -k /dev/stdin -O t %s" %
                       (COUNT, k, fa2, COUNT, k, fa3, sp, TWISTDB, tmp_path / "Classes", tmp_path / "multi"), shell=True, capture_output=True, text=True)
    assert r.returncode == 1 and 'Duplicate_label("c")' in r.stderr, r.stderr  # the text repeats a label the stream brought
    r = subprocess.run("( %s -k %d -L -f %s; %s -k %d -L -f %s ) | %s -I T %s -k /dev/stdin -O t %s" %
                       (COUNT, k, fa2, COUNT, k, fa3, TWISTDB, tmp_path / "Classes", tmp_path / "multi"), shell=True, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = (tmp_path / "multi.KPopTwisted.txt").read_text().splitlines()
    assert [l.split("\t")[0] for l in rows[1:]] == ['"a"', '"b"', '"c"']


def test_several_device_slots_in_one_process(tmp_path, oracle):
    """KPOP_DEVICES: the reads stream cut over device slots inside the library (kpop_init_devices + kpop_sharded_*; here 2
    and 3 slots on the box's one GPU) gives the bytes of the single-device run, for blocks the fused kernel takes and for
    blocks with genomes in them; so do -d and -s; a twister that cannot be loaded is an error, not a hang"""
    rng = np.random.RandomState(77)
    k, d = 12, 40
    make_twister(tmp_path, oracle, k, d, keep=0.02)
    seqs = _reads(rng, n_short=300, genomes=(900, 7000))
    fa = tmp_path / "in.fa"
    write_fasta(fa, [("s%d" % i, s) for i, s in enumerate(seqs)], width=70)
    # binary twister, as the README uses it
    assert subprocess.run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-o", "T", str(tmp_path / "Classes")]).returncode == 0
    outs = {}
    for devices in ("", "2", "3"):
        env = dict(os.environ, KPOP_PIPE_FORMAT="reads", KPOP_SEQ_BLOCK="20000")  # several blocks
        if devices:
            env["KPOP_DEVICES"] = devices
        r = subprocess.run("%s -k %d -L -f %s | %s -v -i T %s -k /dev/stdin -o t %s" % (COUNT, k, fa, TWISTDB, tmp_path / "Classes", tmp_path / ("w" + devices)),
                           shell=True, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert ("device slots" in r.stderr) == bool(devices)
        outs[devices] = (tmp_path / ("w" + devices + ".KPopTwisted")).read_bytes()
    assert outs[""] == outs["2"] == outs["3"] and len(outs[""]) > 1000
    # -d and -s with their second operand's rows cut over the slots: the same files
    for devices in ("", "3"):
        env = dict(os.environ)
        if devices:
            env["KPOP_DEVICES"] = devices
        r = subprocess.run([TWISTDB, "-i", "T", str(tmp_path / "Classes"), "-i", "t", str(tmp_path / "w"), "-d", str(tmp_path / "w"), "-o", "d", str(tmp_path / ("dd" + devices)),
                            "-s", str(tmp_path / "w"), str(tmp_path / ("ss" + devices))], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
    assert (tmp_path / "dd.KPopDMatrix").read_bytes() == (tmp_path / "dd3.KPopDMatrix").read_bytes()
    assert (tmp_path / "ss.KPopSummary.txt").read_bytes() == (tmp_path / "ss3.KPopSummary.txt").read_bytes()
    # a twister that cannot be loaded: the process stops with the message
    (tmp_path / "Bad.KPopTwister").write_bytes(b"not an archive")
    (tmp_path / "Bad.KPopInertia.txt").write_text((tmp_path / "Classes.KPopInertia.txt").read_text())
    r = subprocess.run("%s -k %d -L -f %s | %s -I T %s -k /dev/stdin -o t %s" % (COUNT, k, fa, TWISTDB, tmp_path / "Missing", tmp_path / "x"),
                       shell=True, capture_output=True, text=True, env=dict(os.environ, KPOP_DEVICES="2", KPOP_PIPE_FORMAT="reads"))
    assert r.returncode != 0 and "cannot open" in r.stderr


def test_reads_stream_fastq_and_mates(tmp_path, oracle):
    """single-end and paired-end FASTQ through the reads stream: the same bytes as through text spectra (mates alternate,
    bin/KPopCount.ml:36-54, and share their tag: the consumer's Duplicate_label is the reference's behaviour there)"""
    from test_gpu_cli import write_fastq
    rng = np.random.RandomState(5)
    k, d = 7, 33
    make_twister(tmp_path, oracle, k, d, keep=0.5)
    r1 = [("a%d" % i, "".join(rng.choice(list("ACGTN"), size=int(rng.randint(0, 160))))) for i in range(50)]
    r2 = [("b%d" % i, "".join(rng.choice(list("ACGT"), size=int(rng.randint(0, 160))))) for i in range(50)]
    f1, f2 = tmp_path / "x_1.fastq", tmp_path / "x_2.fastq"
    write_fastq(f1, r1)
    write_fastq(f2, r2)
    outs = {}
    for fmt in ("reads", "text"):
        env = dict(os.environ, KPOP_PIPE_FORMAT=fmt, KPOP_SEQ_BLOCK="2000")
        for name, inputs in (("se", "-s %s -s %s" % (f1, f2)), ("pe", "-p %s %s" % (f1, f2))):
            r = subprocess.run("%s -k %d -L %s | %s -I T %s -k /dev/stdin -o t %s" % (COUNT, k, inputs, TWISTDB, tmp_path / "Classes", tmp_path / (name + fmt)),
                               shell=True, capture_output=True, text=True, env=env)
            assert r.returncode == 0, r.stderr
            outs[name, fmt] = (tmp_path / (name + fmt + ".KPopTwisted")).read_bytes()
    assert outs["se", "reads"] == outs["se", "text"] and outs["pe", "reads"] == outs["pe", "text"]
    assert len(outs["pe", "reads"]) == len(outs["se", "reads"])  # the same 100 sequences either way


def test_reads_stream_empty_and_degenerate_inputs(tmp_path, oracle):
    """an empty FASTA, a FASTA of header-only and too-short records: the reads stream and the text pipeline agree (both leave
    an empty or all-zero register behind, no error)"""
    k, d = 6, 40
    make_twister(tmp_path, oracle, k, d)
    (tmp_path / "empty.fa").write_text("")
    write_fasta(tmp_path / "short.fa", [("only-header", ""), ("tiny", "ACG"), ("allN", "NNNNNNNNNNNN")])
    for name in ("empty", "short"):
        outs = []
        for fmt in ("reads", "text"):
            rc1, rc2, e1, e2 = _pipeline(tmp_path, tmp_path / (name + ".fa"), k, name + fmt, fmt, table=True)
            assert rc1 == 0 and rc2 == 0, (name, fmt, e1, e2)
            outs.append((tmp_path / (name + fmt + ".KPopTwisted.txt")).read_bytes())
        assert outs[0] == outs[1], name
    rows = (tmp_path / "shortreads.KPopTwisted.txt").read_text().splitlines()
    assert len(rows) == 4 and all(set(l.split("\t")[1:]) == {"0"} for l in rows[1:])
