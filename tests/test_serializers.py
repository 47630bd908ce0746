"""CPU tests of the wire formats (reference plonk_serializers.py:23-289):
round trips through JSON for every type, with values taken from the golden fixtures."""
import json
import os

from zkhip.field import FQ, FQ2, FR
from zkhip.plonk.polynomial import Polynomial
from zkhip.plonk.preprocessor import PreprocessedData
from zkhip.plonk.prover import Proof
from zkhip.plonk.srs import SRS
from zkhip.plonk.transcript import Transcript
from zkhip import serializers as ser


def test_scalars_points_polys_roundtrip(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "kzg_seed42.json")))
    toy = json.load(open(os.path.join(golden_dir, "toy_groth16.json")))
    x = FR(int(g["tau"]))
    assert ser.serialize_fr(x) == g["tau"] and ser.deserialize_fr(g["tau"]) == x
    p = ser.deserialize_g1(g["g1_powers"][3])
    assert isinstance(p[0], FQ) and ser.serialize_g1(p) == g["g1_powers"][3]
    assert ser.serialize_g1(None) is None and ser.deserialize_g1(None) is None
    q = ser.deserialize_g2(toy["proof_B"])
    assert isinstance(q[0], FQ2) and ser.serialize_g2(q) == toy["proof_B"] and ser.deserialize_g2(None) is None
    poly = Polynomial([FR(1), FR(0), FR(5)])
    assert ser.serialize_poly(poly) == ["1", "0", "5"] and ser.deserialize_poly(["1", "0", "5"]) == poly
    assert ser.deserialize_fr_list(ser.serialize_fr_list([FR(3), FR(4)])) == [FR(3), FR(4)]
    assert ser.g1_short(None) == "∞" and ser.g1_short((FQ(1), FQ(2))) == "(1, 2)"
    from zkhip.field import G2
    assert ser.g2_short(None) == "∞" and ser.g2_short(G2) == "(1085...2781+1155...5634i, ...)"
    assert ser.fr_short(None) == "None" and ser.fr_short(FR(1234567890)) == "1234567890" and ser.fr_short(FR(12345678901)) == "1234...8901"
    assert ser.g1_from_ints(ser.turn_g1_int(p)) == p and ser.g2_from_ints(ser.turn_g2_int(q)) == q


def test_srs_transcript_proof_preprocessed_roundtrip(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "kzg_seed42.json")))
    srs = SRS([ser.deserialize_g1(p) for p in g["g1_powers"]], [ser.deserialize_g2(p) for p in g["g2_powers"]], 8)
    back = ser.deserialize_srs(json.loads(json.dumps(ser.serialize_srs(srs))))
    assert back.g1_powers == srs.g1_powers and back.g2_powers == srs.g2_powers and back.max_degree == 8
    t = Transcript()
    t.append_point(b"a_comm", srs.g1_powers[1])
    t.append_scalar(b"x", FR(9))
    t2 = ser.deserialize_transcript(ser.serialize_transcript(t))
    assert t2.challenge_scalar(b"beta") == t.challenge_scalar(b"beta")
    proof = Proof()
    for i, name in enumerate(Proof.FIELDS):
        setattr(proof, name, srs.g1_powers[i % 9] if name.endswith("comm") else FR(1000 + i))
    proof.t_hi_comm = None
    pb = ser.deserialize_proof(json.loads(json.dumps(ser.serialize_proof(proof))))
    assert all(getattr(pb, f) == getattr(proof, f) for f in Proof.FIELDS)
    pp = PreprocessedData()
    pp.n, pp.omega, pp.domain, pp.sigma, pp.num_public_inputs = 4, FR(7), [FR(1), FR(7), FR(49), FR(343)], list(range(12)), 1
    for k, name in enumerate(("q_l", "q_r", "q_o", "q_m", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")):
        setattr(pp, name + "_poly", Polynomial([FR(k), FR(k + 1)]))
        setattr(pp, name + "_comm", srs.g1_powers[k])
    qq = ser.deserialize_preprocessed(json.loads(json.dumps(ser.serialize_preprocessed(pp))))
    assert qq.n == 4 and qq.omega == FR(7) and qq.domain == pp.domain and qq.sigma == pp.sigma
    assert qq.q_m_poly == pp.q_m_poly and qq.s_sigma3_comm == pp.s_sigma3_comm
