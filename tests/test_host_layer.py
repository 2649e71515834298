"""CPU tests of the host layer (no GPU): config/input parsing, Predict/Main plumbing and the
printed text, byte-for-byte against the reference's outputs (tests/golden/text/*.txt).
The compute answers come from the CPU oracle through the test-only OracleEngine."""
import hashlib
import io
import json
import os

import pytest

from squarna_amd import engine as E
from tests.oracle_engine import OracleEngine

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATA = os.path.join(os.path.dirname(GOLDEN), "..", "squarna_amd", "data")

with open(os.path.join(GOLDEN, "digests.json")) as f:
    DIGESTS = json.load(f)

FAST = ["s16_nobpp", "s16_fastest", "shape_input_fastest", "shape_input_alt_rf26", "seq_input_evalonly",
        "seq_input_entropy", "seq_input_ico", "seq_input_nobpp", "seq_input_greedynobpp_rf10",
        "SRtest150_fastest", "SRtest150_fastest_pl1", "ali_input_a", "ali_input_a_verbose", "ali_input_a_s3i",
        "ali_input_a_s31", "demo_afa_a"]


def run_predict(args, engine):
    from squarna_amd import Predict
    kw = dict(args)
    if "inputfile" in kw:
        kw["inputfile"] = os.path.join(DATA, kw["inputfile"])
    buf = io.StringIO()
    with E.use_engine(engine):
        Predict(write_to=buf, **kw)
    return buf.getvalue()


@pytest.mark.parametrize("tag", FAST)
def test_predict_text_matches_reference(tag):
    txt = run_predict(DIGESTS[tag]["args"], OracleEngine())
    with open(os.path.join(GOLDEN, "text", tag + ".txt")) as f:
        exp = f.read()
    assert txt == exp
    assert hashlib.sha256(txt.encode()).hexdigest() == DIGESTS[tag]["sha256"]


def test_parse_config_all_shipped():
    from squarna_amd.config import ParseConfig, builtin_config
    for name in ("def", "alt", "fastest", "nobpp", "greedynobpp", "500", "1000", "ali", "500nobpp",
                 "1000nobpp", "edmondsnobpp", "hungariannobpp", "nussinovnobpp", "greedy", "edmonds",
                 "hungarian", "nussinov"):
        names, psets = ParseConfig(builtin_config(name))
        assert len(names) == len(psets) >= 1
        assert all("bpweights" in p and "algorithms" in p for p in psets)
    names, psets = ParseConfig(builtin_config("nobpp"))
    assert names == ["defG1", "defG2", "defN", "defE", "defH"]
    assert psets[1]["bpweights"] == {"GC": 2.0, "AU": 1.0, "GU": 1.0} and psets[1]["minlen"] == 2.0


def test_validation_messages():
    from squarna_amd import Predict
    with pytest.raises(AssertionError, match="Input file does not exist"):
        Predict(inputfile="/nonexistent/file.fas")
    with pytest.raises(ValueError, match="Inappropriate toplim value"):
        Predict(inputseq="ACGU", configfile="fastest", toplim="x")
    with pytest.raises(AssertionError, match="Inappropriate rankby value"):
        Predict(inputseq="ACGU", configfile="fastest", rankby="q")
    with pytest.raises(AssertionError, match="Config file does not exist"):
        Predict(inputseq="ACGU", configfile="nope")
    with pytest.raises(NotImplementedError):
        Predict(inputseq="ACGU", configfile="fastest", rfam=True)


def test_main_cli_forms(capsys, monkeypatch):
    from squarna_amd import api
    with E.use_engine(OracleEngine()):
        monkeypatch.setattr("sys.argv", ["SQUARNA", "s=ACGUACGUACUCGACG", "c=nobpp"])
        api.Main()
        a = capsys.readouterr().out
        monkeypatch.setattr("sys.argv", ["SQUARNA", "-s", "ACGUACGUACUCGACG", "--config", "nobpp"])
        api.Main()
        b = capsys.readouterr().out
    assert a == b
    with open(os.path.join(GOLDEN, "text", "s16_nobpp.txt")) as f:
        assert a == "None\n" + f.read()


def test_no_cpu_fallback_without_gpu():
    """The default engine must fail loudly when there is no GPU (no silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from squarna_amd import Predict
    with pytest.raises(RuntimeError, match="no CPU fallback|libsquarna_hip"):
        Predict(inputseq="ACGUACGUACUCGACG", configfile="fastest", write_to=io.StringIO())


def test_c_abi_exports_every_declared_symbol():
    """libsquarna_hip.so loads and exports every function include/squarna_hip.h declares."""
    import re
    from squarna_amd import _lib
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "squarna_hip.h")).read()
    declared = set(re.findall(r"SQ_API\s+[\w\s\*]+?\b(sq_\w+)\s*\(", hdr))
    assert declared, "no declarations found"
    L = _lib.load()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(_lib.SYMBOLS) == declared
    assert L.sq_version() >= 100


def test_string_and_pair_helpers_match_reference_goldens():
    """PairsToDBN (incl. levellimit / returnlevels), DBNToPairs, UnAlign, ReAlign, PairsToStems of the host layer
    against values the reference returned (tests/golden/gen_helpers_golden.py)."""
    import json
    import os
    from squarna_amd.dbn import DBNToPairs, PairsToDBN, PairsToStems, ReAlign, UnAlign
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "helpers.json")) as f:
        g = json.load(f)
    for c in g["pairs_to_dbn"]:
        pairs = [tuple(p) for p in c["pairs"]]
        assert PairsToDBN(list(pairs), c["length"], levellimit=c["levellimit"]) == c["dbn"], c
        lev = PairsToDBN(list(pairs), c["length"], returnlevels=True)
        assert sorted([v, w, l] for (v, w), l in lev.items()) == c["levels"], c
    for c in g["pairs_to_stems"]:
        got = PairsToStems([tuple(p) for p in c["pairs"]])
        assert [[[list(bp) for bp in st[0]], st[1]] for st in got] == c["stems"], c
    for c in g["dbn_to_pairs"]:
        assert [list(p) for p in DBNToPairs(c["dbn"])] == c["pairs"], c
    for c in g["unalign"]:
        assert list(UnAlign(c["seq"], c["dbn"])) == c["out"], c
    for c in g["realign"]:
        assert ReAlign(c["shortdbn"], c["longseq"]) == c["out"], c
        short_seq = UnAlign(c["longseq"], "." * len(c["longseq"]))[0]
        assert ReAlign(short_seq, c["longseq"], seqmode=True) == c["out_seqmode"], c
