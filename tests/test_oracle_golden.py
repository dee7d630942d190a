"""CPU tier: the oracle (our CPU restatement) against the committed golden vectors that the REAL
reference produced (tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import numpy as np
import pytest

from util import golden_cases, load_case, load_kat


def test_kat_rows(oracle):
    kat = load_kat()
    seq = kat["seq"]
    assert len(kat["rows"]) >= 20
    for row in kat["rows"]:
        k, m, H = row["k"], row["m"], row["H"]
        s = seq[row["start"]:row["start"] + k]
        f, r = oracle.encode_fwd(s), oracle.encode_rc(s)
        assert [int(x) for x in f] == row["fwd"]
        assert [int(x) for x in r] == row["rc"]
        assert [int(x) for x in oracle.complement(f, k)] == row["comp"]
        cn = oracle.canonical(f, r, k)
        assert [int(x) for x in cn] == row["canon"]
        assert oracle.std_hash(cn, k) == row["std_hash"]
        assert oracle.double_hash(cn, k) == (row["h1"], row["h2"])
        assert [int(x) for x in oracle.probe_positions(row["h1"], row["h2"], m, H)] == row["pos"]
        assert oracle.decode(f, k) == s


def test_kat_survey_table(oracle):
    # SURVEY.md section 8c, row 2 and the extra probe-confirmed answers
    assert oracle.std_hash(np.array([0], np.uint64), 31) == 0xf457bd22cac02869
    assert oracle.std_hash(np.array([0x123456789], np.uint64), 21) == 0x6b77ebaa916f5a5b
    key = np.array([0x0123456789abcdef], np.uint64)
    assert oracle.std_hash(key, 31) == 0x8428d319057264f0
    assert oracle.double_hash(key, 31) == (0xcfdda4a6741cf403, 0x17fc9bd667b31253)
    assert [int(x) for x in oracle.probe_positions(0xcfdda4a6741cf403, 0x17fc9bd667b31253, 66855782, 3)] == \
        [45683435, 10017376, 41207099]
    s = "ACGTACGTACGTACGTACGTACGTACGTACG"
    assert int(oracle.encode_fwd(s)[0]) == 0x06c6c6c6c6c6c6c6
    assert int(oracle.encode_rc(s)[0]) == 0x1b1b1b1b1b1b1b1b


def test_kat_hash_words(oracle):
    for blk in load_kat()["hash_words"]:
        k = blk["k"]
        for w, sh, h1, h2 in zip(blk["words"], blk["std_hash"], blk["h1"], blk["h2"]):
            key = np.array(w, np.uint64)
            assert oracle.std_hash(key, k) == sh, (k, w)
            assert oracle.double_hash(key, k) == (h1, h2)


def test_unknown_characters(oracle):
    u = load_kat()["unknown_chars"]
    s = u["s"]  # 22 chars into a 21-base key: the first char is shifted out
    # E1 over the whole string == rolling all 22 characters through a 42-bit key
    from oracle.p3_oracle import _flat  # noqa: F401
    f = 0
    for ch in s:
        f = ((f << 2) | int(oracle.encode_fwd(ch + "A" * 20)[0] >> 40)) & ((1 << 42) - 1)
    assert f == u["fwd"]
    assert oracle.decode(np.array([u["fwd"]], np.uint64), 21) == u["dec_fwd"] == "CGAAAAAAACGTACGTACGTA"
    assert oracle.decode(np.array([u["rc"]], np.uint64), 21) == u["dec_rc"] == "ACGTACGTACGTAAAAAACGT"
    # E2 on the last 21 characters gives the same rc key only if no base was shifted out; check E2 directly
    # on a 21-char string with unknowns instead
    t = s[1:]
    assert oracle.decode(oracle.encode_fwd(t), 21) == "CGAAAAAAACGTACGTACGTA"


def test_rmq(oracle):
    for row in load_kat()["rmq"]:
        assert [int(x) for x in oracle.rmq(row["v"], row["x"])] == row["out"]


def test_estimate_bloom(oracle):
    for row in load_kat()["estimate_bloom"]:
        assert oracle.estimate_bloom(row["all_bases"], row["k"]) == (row["m"], row["H"])
    assert oracle.estimate_bloom(150000000, 31, 1 << 30) == (1 << 30, 10)


@pytest.mark.parametrize("name", golden_cases())
def test_stage_golden(oracle, name):
    c = load_case(name)
    keys, counts = oracle.count_short(c["reads"], min_len=c["k"])
    assert np.array_equal(keys, c["keys"])
    assert np.array_equal(counts, c["counts"])
    words, seeds, nins, ovf = oracle.make_bf(c["reads"], keys, counts, c["k"], c["m"], c["H"])
    assert np.array_equal(words, c["bloom"])
    assert seeds == c["seeds"]
    assert ovf == 0
    # B2: every canonical k-mer that was inserted must query positive; refshaped mode agrees too
    rs = oracle.refshaped_pass(c["reads"], c["k"], c["m"], c["H"], want_bloom=True)
    assert np.array_equal(rs["bloom"], c["bloom"])
    assert rs["n_seeds"] == len(c["seeds"]) and rs["n_distinct"] == c["keys"].size
