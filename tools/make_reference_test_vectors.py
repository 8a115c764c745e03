"""Collects the known-answer DATA (input sequences + expected numbers) that the reference's own unit tests hold
for the DSK hot path into tests/golden/reference_unit_vectors.json.

Only data is extracted: nucleotide strings and expected integers. No reference code is copied.
Sources (under /root/reference/gatb-core/test/unit/src):
  kmer/TestDSK.cpp:147-241   DSK_check1   (solid counts for (sequences, k, nks))
  kmer/TestDSK.cpp:254-305   DSK_check2   (exact k=31 solid k-mers + checksum)
  kmer/TestDSK.cpp:482-612   DSK_perBank1/2 (sum-solidity rows only: single-bank semantics)
  kmer/TestKmer.cpp:141-190  direct / canonical k-mer values
  kmer/TestKmer.cpp:468-489  minimizer table k=15 m=7
  kmer/TestKmer.cpp:551-561  bad-char validity table
  tools/math/TestMath.cpp:94-98  revcomp / simplehash16 / hash1 / oahash constants
  kmer/TestDebloom.cpp:84-137    20 critical false positives (pins BLOOM_BASIC bit positions)
"""
import json, re, sys

R = "/root/reference/gatb-core/test/unit/src/"


def literals(block):
    """concatenate adjacent C string literals; split on commas between them"""
    seqs, cur = [], []
    for line in block.splitlines():
        line = line.split("//")[0].strip()
        m = re.findall(r'"([ACGTN]*)"', line)
        if m:
            cur.append("".join(m))
        if line.endswith(",") or line.endswith("};") or line.endswith("} ;"):
            if cur:
                seqs.append("".join(cur)); cur = []
    if cur:
        seqs.append("".join(cur))
    return seqs


dsk = open(R + "kmer/TestDSK.cpp").read()
i = dsk.index("const char* seqs4[] = {"); j = dsk.index("} ;", i)
seqs4 = literals(dsk[i:j + 3])
assert len(seqs4) == 3, len(seqs4)
s1 = re.search(r'const char\* s1 = "([ACGT]+)" ;\s*\n\s*const char\* seqs1', dsk).group(1)
check1 = []
for m in re.finditer(r"DSK_check1_aux \((seqs\d), ARRAY_SIZE\(seqs\d\),\s*(\d+),\s*(\d+),\s*(\d+)\);", dsk):
    check1.append([m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))])
i = dsk.index("void DSK_check2_aux"); blk = dsk[i:i + 3000]
s2 = re.search(r'const char\* s1 = "([ACGT]+)"', blk).group(1)
vals2 = [int(x, 16) for x in re.findall(r"setVal\(\s*(0x[0-9A-Fa-f]+)\s*\)", blk)]
cs2 = int(re.search(r"checksum == (0x[0-9a-fA-F]+)", blk).group(1), 16)

i = dsk.index("void DSK_perBank1"); blk = dsk[i:dsk.index("void DSK_perBank2")]
pb1_seqs = re.findall(r'"([ACGT]+)",?\s*//', blk)
pb1 = [[int(a), int(b)] for a, b in re.findall(r"album, 15, (\d+), nksMax, KMER_SOLIDITY_SUM, (\d+)\)", blk)]
i = dsk.index("void DSK_perBank2"); blk = dsk[i:dsk.index("void DSK_perBankKmer_aux")]
pb2_seqs = re.findall(r'"([ACGT]+)",?\s*//', blk)[:3]
pb2 = [[int(a), (2 ** 30 if b == "nksMax" else int(b)), int(c)]
       for a, b, c in re.findall(r"album, 5, (\d+), (\w+), KMER_SOLIDITY_SUM, (\d+)\)", blk)]

km = open(R + "kmer/TestKmer.cpp").read()
seq_k3 = re.search(r'const char\* seq = "(CATTGATAGTGG)"', km).group(1)
direct = [int(x) for x in re.search(r"long checkDirect \[\]\s*=\s*\{([^}]*)\}", km).group(1).split(",")]
both = [int(x) for x in re.search(r"long checkBoth \[\]\s*=\s*\{([^}]*)\}", km).group(1).split(",")]
i = km.index("void kmer_minimizer3"); blk = km[i:i + 3000]
mini_seq = re.search(r'const char\* seq = "([ACGT]+)"', blk).group(1)
mini_tab = [[a, b, int(c), d == "true"] for a, b, c, d in
            re.findall(r'\{"([ACGT]+)",\s*"([ACGT]+)",\s*(\d+),\s*(true|false)\s*\}', blk)]
i = km.index("void kmer_badchar"); blk = km[i:i + 3000]
bad_seq = re.search(r'const char\* seq = "([ACGTN]+)"', blk).group(1)
bad_tab = [[a, b == "true"] for a, b in re.findall(r'\{"([ACGTN]+)",\s*(true|false)\s*\}', blk)]

mt = open(R + "tools/math/TestMath.cpp").read()
math = dict(
    revcomp=[int(x, 16) if x.startswith("0x") else int(x) for x in
             re.search(r"revcomp \(CST \((0x[0-9a-f]+)\), (\d+)\) ==\s+CST \((0x[0-9a-f]+)\)", mt).groups()],
    simplehash16=[int(x, 16) if x.startswith("0x") else int(x) for x in
                  re.search(r"simplehash16 \(CST\((0x[0-9a-f]+)\), (\d+)\) == (\d+)", mt).groups()],
    hash1=[int(x, 16) if x.startswith("0x") else int(x) for x in
           re.search(r"hash1\(CST\((0x[0-9a-f]+)\), (\d+)\)\s+==\s+(\d+)UL", mt).groups()],
    oahash=[int(x, 16) if x.startswith("0x") else int(x) for x in
            re.search(r"oahash\(CST\((0x[0-9a-f]+)\)\)\s+== (\d+)", mt).groups()],
)

db = open(R + "kmer/TestDebloom.cpp").read()
i = db.index("void Debloom_check1"); blk = db[i:i + 4000]
db_seq = "".join(re.findall(r'"([ACGT]+)"', blk[:blk.index("} ;")]))
db_vals = [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", blk[blk.index("u_int64_t values[]"):blk.index("};", blk.index("u_int64_t values[]"))])]
assert len(db_vals) == 20

out = dict(
    dsk_check1=dict(seqs1=[s1], seqs2=[s1, s1], seqs3=[s1, s1, s1], seqs4=seqs4, cases=check1),
    dsk_check2=dict(seq=s2, k=31, values=vals2, checksum=cs2),
    dsk_perbank1=dict(seqs=pb1_seqs, k=15, sum_cases=pb1),
    dsk_perbank2=dict(seqs=pb2_seqs, k=5, sum_cases=pb2),
    kmer_k3=dict(seq=seq_k3, direct=direct, canonical=both),
    minimizer_k15_m7=dict(seq=mini_seq, k=15, m=7, table=mini_tab),
    badchar_k11=dict(seq=bad_seq, k=11, table=bad_tab),
    math=math,
    debloom_k11=dict(seq=db_seq, k=11, m=8, cfp=db_vals),
)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/reference_unit_vectors.json", "w"), indent=1)
print({k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})
print(len(check1), vals2, hex(cs2), pb1, pb2[:4], len(pb2), math, len(mini_tab), len(bad_tab), len(db_seq))
