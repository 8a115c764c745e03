"""Device FASTA / FASTQ parser (gkc_fastx_parse_device, gkc_push_fastx) against the oracle's restatement of the reference reader
(gko.fastx_parse = BankFasta.cpp:488-571) — byte-exact bases and offsets — on the reference's own bank fixtures and on generated text."""
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads

pytestmark = pytest.mark.gpu
BANK = os.path.join(os.path.dirname(__file__), "golden", "bank")


@pytest.fixture(scope="module")
def gkc():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return ge.load().gkc


def check(gkc, text):
    c = gkc.Counter(0)
    b, o, cons = c.fastx_parse(text)
    rb, ro = gko.fastx_parse(text)
    assert np.array_equal(o, ro), (len(o), len(ro))
    assert np.array_equal(b, rb)
    assert cons == len(text)
    return b, o


@pytest.mark.parametrize("name", ["sample1.fa", "sample2.fa", "sample.fastq", "reads1.fa"])
def test_reference_fixtures(gkc, name):
    b, o = check(gkc, open(os.path.join(BANK, name), "rb").read())
    assert len(o) - 1 == {"sample1.fa": 20, "sample2.fa": 20, "sample.fastq": 7, "reads1.fa": 100}[name]    # TestBank.cpp known answers


def fasta_text(reads, width, rng, crlf=False, blank=False, final_nl=True):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i, r in enumerate(reads):
        out.append(b">r%d some comment" % i + eol)
        w = width if width else int(rng.integers(1, 90))
        for j in range(0, len(r), w):
            out.append(r[j:j + w] + eol)
            if blank and rng.random() < 0.1:
                out.append(eol)
    t = b"".join(out)
    return t if final_nl else t.rstrip(b"\r\n")


def fastq_text(reads, rng, crlf=False, final_nl=True):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i, r in enumerate(reads):
        q = bytes(rng.integers(33, 74, size=len(r)).astype(np.uint8))      # may start with '@' or '+' or '>'
        out += [b"@r%d/1" % i + eol, r + eol, b"+" + (b"r%d/1" % i if i % 3 == 0 else b"") + eol, q + eol]
    t = b"".join(out)
    return t if final_nl else t.rstrip(b"\r\n")


@pytest.mark.parametrize("width,crlf,blank,final_nl", [(70, False, False, True), (0, False, True, True), (60, True, False, True), (0, True, True, False), (1, False, False, True)])
def test_fasta_generated(gkc, width, crlf, blank, final_nl):
    rng = np.random.default_rng(width + 10 * crlf)
    reads = [bytes(r) for r in synth_reads(3000 if width != 1 else 300, 20000, 150, seed=3, n_rate=0.01, ragged=True)]
    check(gkc, fasta_text(reads, width, rng, crlf, blank, final_nl))


@pytest.mark.parametrize("crlf,final_nl", [(False, True), (True, True), (False, False)])
def test_fastq_generated(gkc, crlf, final_nl):
    rng = np.random.default_rng(5)
    reads = [bytes(r) for r in synth_reads(20000, 100000, 150, seed=4, n_rate=0.01, ragged=True)]
    check(gkc, fastq_text(reads, rng, crlf, final_nl))


def test_edge_texts(gkc):
    for t in [b"", b"\n\n", b">", b">x", b">a\r\n\r\nAC\r\n", b">a\nAC\n\n\nGT", b"junk\n\n>a\nAC\n", b">a\n>b\n>c\nA\n", b"@r\nA\n+\nI", b"@r\n\n+\n\n@s\nAC\n+\nII\n",
              b">a\nAC GT\n", b">only header\n", b"no header at all\nACGT\n"]:
        check(gkc, t)


def test_refused_texts(gkc):
    c = gkc.Counter(0)
    for t in [b"@r1\nACGT\n+\nII\n@r2\nGG\n+\nII\n",            # quality shorter than the sequence
              b"@r1\nAC\nGT\n+\nIIII\n@r2\nGG\n+\nII\n",       # multi-line FASTQ
              b">a\nAC\n+\nII\n>b\nAC\nGT\nAA\n",              # '+' line in a FASTA-shaped file (first record decides: here FASTQ, then broken)
              b">a\nACGT\nAC\n+AC\n",                          # sequence line starting with '+' in FASTA
              b"x>y\n>a\nAC\n"]:                                # '>' before the first header line
        with pytest.raises(gkc.GkcError):
            c.fastx_parse(t)


def _fuzz_text(rng, fastq):
    """well-formed FASTA (multi-line, blank lines, CRLF, odd header text) or 4-line FASTQ (quality strings full of '@', '>' and '+')"""
    nl = b"\r\n" if rng.random() < 0.3 else b"\n"
    out = []
    alpha = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)
    qual = np.frombuffer(b"@>+!IJ#;~ABCDEFGH", dtype=np.uint8)
    head = np.frombuffer(b"abc XYZ_0123|:>@+ \t", dtype=np.uint8)
    for _ in range(int(rng.integers(1, 12))):
        h = rng.choice(head, int(rng.integers(0, 20))).tobytes().replace(b"\r", b"")
        L = int(rng.integers(0, 200))
        seq = rng.choice(alpha, L).tobytes()
        if fastq:
            out += [b"@" + h, seq, b"+" + (h if rng.random() < 0.5 else b""), rng.choice(qual, L).tobytes()]
        else:
            out.append(b">" + h)
            w = int(rng.integers(1, 90))
            lines = [seq[i:i + w] for i in range(0, L, w)] if rng.random() < 0.7 else [seq]
            for ln in lines:
                out.append(ln)
                if rng.random() < 0.1:
                    out.append(b"")
    text = nl.join(out)
    if rng.random() < 0.8:
        text += nl
    return text


def test_fuzz_wellformed_texts(gkc):
    rng = np.random.default_rng(2024)
    for it in range(300):
        t = _fuzz_text(rng, fastq=bool(it & 1))
        check(gkc, t)


def test_chunked_feed_equals_whole(gkc):
    rng = np.random.default_rng(9)
    reads = [bytes(r) for r in synth_reads(5000, 50000, 150, seed=8, ragged=True)]
    for text in (fasta_text(reads, 60, rng), fastq_text(reads, rng)):
        rb, ro = gko.fastx_parse(text)
        c = gkc.Counter(0)
        pos, carry, got_b, got_len = 0, b"", [], []
        step = 100_003
        while pos < len(text) or carry:
            chunk = carry + text[pos:pos + step]; pos += step
            final = pos >= len(text)
            b, o, cons = c.fastx_parse(chunk, final=final)
            got_b.append(b); got_len.append(np.diff(o))
            carry = chunk[cons:]
            if final:
                assert cons == len(chunk); break
        assert np.array_equal(np.concatenate(got_b), rb)
        assert np.array_equal(np.concatenate(got_len), np.diff(ro))


def test_push_fastx_counts_like_push_reads(gkc):
    k, m, parts = 31, 10, 16
    rng = np.random.default_rng(2)
    reads = [bytes(r) for r in synth_reads(4000, 30000, 150, seed=6, n_rate=0.002, ragged=True)]
    rep = simple_repart(m, parts)
    res = []
    for mode in ("reads", "fasta", "fastq"):
        c = gkc.Counter(0); c.configure(k, m, parts, rep); c.begin_pass(0)
        if mode == "reads":
            c.push_reads(*gko.pack_reads(reads))
        else:
            t = fasta_text(reads, 70, rng) if mode == "fasta" else fastq_text(reads, rng)
            assert c.push_fastx(t) == len(t)
        c.finish_pass()
        res.append((c.stats()["kmers_nb_distinct"], c.stats()["kmers_nb_valid"], c.result_checksum()))
    assert res[0] == res[1] == res[2]
