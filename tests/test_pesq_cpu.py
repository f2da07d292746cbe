"""CPU: the PESQ oracle (oracle/pesq_ref.py, oracle/pesq_tables.py).  pesq==0.0.4 is absent from the image, so the oracle is
"parity unpinned" against the package; what CAN be checked without it is checked here: the redundancy of the standard's Bark
tables, the known ceilings of the two MOS mappings, monotonicity, recovery of known delays, and the committed regression
vectors."""
import json
import os

import numpy as np
import pytest

from oracle import pesq_ref, pesq_tables
from tests import pesq_cases

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pesq_oracle.npz"))


def test_bark_tables_are_self_consistent():
    """centre = cumulative widths, correction = width_hz / (width_bark * bins), bins fall between the Hz edges, 128 bins, 4 kHz:
    what pins the restated 8 kHz tables digit for digit."""
    assert pesq_tables.check_redundancy()
    t16 = pesq_tables.tables(16000)
    assert t16["nb"] == 49 and int(t16["nr"].sum()) == 256 and abs(t16["width_hz"].sum() - 8000.0) < 1.0
    assert np.array_equal(t16["nr"][:41], pesq_tables.tables(8000)["nr"][:41])
    assert np.all(np.diff(t16["centre_bark"]) > 0)


def test_generated_kernel_tables_match_the_oracle_tables():
    """csrc/pesq_tables.h is generated from oracle/pesq_tables.py: the committed header must be the current one."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "urgent2026_challenge_track1_amd", "csrc", "pesq_tables.h")
    before = open(path).read()
    subprocess.run([sys.executable, os.path.join(root, "scripts", "gen_pesq_tables.py")], check=True, capture_output=True)
    assert open(path).read() == before


@pytest.mark.parametrize("i", [0, 4, 5, 8, 10])
def test_oracle_regression_vectors(i):
    fs, mode, ref, deg = pesq_cases.make_case(i)
    mos, tr = pesq_ref.pesq(fs, ref, deg, mode, return_trace=True)
    want = json.loads(str(GOLD["trace"][i]))
    got = json.loads(json.dumps({k: tr.get(k) for k in pesq_cases.TRACE_KEYS}, default=lambda o: o.tolist() if hasattr(o, "tolist") else float(o)))
    assert got == want
    if i == 10:
        assert mos == pesq_ref.NO_UTTERANCES_DETECTED
    else:
        assert abs(mos - float(GOLD["mos"][i])) <= 1e-6


def test_identical_signals_reach_the_mapping_ceilings():
    """ref == deg -> raw 4.5 -> 4.5486 (P.862.1) / 4.6439 (P.862.2): the published maxima of the two mappings."""
    for i, top in ((0, 4.5486), (5, 4.6439)):
        fs, mode, ref, _ = pesq_cases.make_case(i)
        assert abs(pesq_ref.pesq(fs, ref, ref, mode) - top) < 1e-3


def test_delay_is_recovered_and_score_falls_with_noise():
    fs, mode, ref, deg = pesq_cases.make_case(6)           # 37-sample delay at 5 dB
    _, tr = pesq_ref.pesq(fs, ref, deg, mode, return_trace=True)
    assert tr["utt_delay"] == [37]
    fs, mode, ref, deg = pesq_cases.make_case(8)           # 20 ms jump in the middle -> two utterances, 0 and 320 samples
    _, tr = pesq_ref.pesq(fs, ref, deg, mode, return_trace=True)
    assert tr["n_utterances"] == 2 and tr["utt_delay"] == [0, 320]
    assert float(GOLD["mos"][1]) > float(GOLD["mos"][2]) and float(GOLD["mos"][5]) > float(GOLD["mos"][6])


@pytest.mark.parametrize("i", [3, 7, 8, 9])
def test_f32_storage_variant_makes_the_same_integer_decisions(i):
    """the ITU code keeps its buffers in C floats; the oracle computes in float64.  With every stored buffer rounded to float32
    (pesq_ref.q) the integer outputs of the alignment stages stay the same (pause, negative delay, utterance split, bad
    interval) and the MOS moves by < 1e-6: the integer-stage parity claim does not hinge on the oracle's precision."""
    fs, mode, ref, deg = pesq_cases.make_case(i)
    mos, tr = pesq_ref.pesq(fs, ref, deg, mode, return_trace=True, precision="f32")
    want = json.loads(str(GOLD["trace_f32"][i]))
    got = json.loads(json.dumps({k: tr.get(k) for k in pesq_cases.TRACE_KEYS}, default=lambda o: o.tolist() if hasattr(o, "tolist") else float(o)))
    assert got == want == json.loads(str(GOLD["trace"][i]))
    assert abs(mos - float(GOLD["mos"][i])) <= 1e-6 and abs(mos - float(GOLD["mos_f32"][i])) <= 1e-9
