"""The oracle's FASTA / FASTQ reader (oracle/gkc_oracle.c gko_fastx_parse, a restatement of BankFasta.cpp:488-571) against the known
answers of the reference's own bank tests (test/unit/src/bank/TestBank.cpp) on the reference's own fixture files
(tests/golden/bank/*: data files of the reference's test/db)."""
import os

import numpy as np

from oracle import gko

BANK = os.path.join(os.path.dirname(__file__), "golden", "bank")


def parse(name):
    return gko.fastx_parse(open(os.path.join(BANK, name), "rb").read())


def seqs(data, offs):
    return [bytes(data[int(offs[i]):int(offs[i + 1])]) for i in range(len(offs) - 1)]


def test_sample1_rotations():
    # TestBank.cpp:98-135 bank_checkSample1_aux: 20 sequences of 20 letters, sequence i = "ARNDCQEGHILKMFPSTWYV" rotated left by i
    text = b"ARNDCQEGHILKMFPSTWYV"
    s = seqs(*parse("sample1.fa"))
    assert len(s) == 20
    for i, x in enumerate(s):
        assert len(x) == 20 and x == bytes(text[(i + k) % 20] for k in range(20))


def test_sample2_empty_sequences():
    # TestBank.cpp:170-200 bank_checkSample2_aux: 20 sequences, every data size 0
    s = seqs(*parse("sample2.fa"))
    assert len(s) == 20 and all(len(x) == 0 for x in s)


def test_sample_fastq():
    # TestBank.cpp:218-243 bank_checkSample3_aux: 7 sequences, every data size > 0
    s = seqs(*parse("sample.fastq"))
    assert len(s) == 7 and all(len(x) > 0 for x in s)


def test_reads1_count():
    # TestBank.cpp:911,938,969: reads1.fa has 100 sequences
    s = seqs(*parse("reads1.fa"))
    assert len(s) == 100
    assert all(set(x) <= set(b"ACGTN") for x in s)


def test_reader_rules():
    f = lambda t: seqs(*gko.fastx_parse(t))
    assert f(b">a\nAC\nGT\n>b c d\nTT\n") == [b"ACGT", b"TT"]                       # multi-line FASTA, header with comment
    assert f(b">a\r\nAC\r\nGT\r\n") == [b"ACGT"]                                   # CRLF: one trailing \r per line dropped (BankFasta.cpp:479)
    assert f(b">a\r\n\r\nAC\r\n") == [b"\rAC"]                                     # ... but not when the accumulated read has length 1
    assert f(b">a\nAC\n\n\nGT") == [b"ACGT"]                                       # empty lines, no final newline
    assert f(b"junk\n>a\nAC\n") == [b"AC"]                                         # everything before the first header is skipped
    assert f(b"@r1\nACGT\n+\nIIII\n@r2\nGG\n+r2\n@I\n") == [b"ACGT", b"GG"]        # quality may start with '@': consumed by length
    assert f(b"@r1\nACGT\n+\nII\nII\n@r2\nGG\n+\nII\n") == [b"ACGT", b"GG"]        # multi-line quality, consumed by length
    assert f(b"@r1\nACGT\n+\nII\n@r2\nGG\n+\nII\n") == [b"ACGT"]                   # short quality swallows the next header line (reference behaviour)
    assert f(b">a\nAC GT\n") == [b"AC GT"]                                         # blanks inside a sequence line are kept
    assert f(b"") == [] and f(b"\n\n") == [] and f(b">") == [] and f(b">x") == [b""]
