"""Wire / on-disk formats of the reference (plonk_serializers.py:23-289, app.py:967-1001) for the
backend's value types, plus the limb-array views the C ABI consumes.

JSON side (what the reference stores in TinyDB): FR -> decimal string; G1 -> [x, y] decimal strings
or None; G2 -> [[x.c0, x.c1], [y.c0, y.c1]]; polynomial -> list of coefficient strings; transcript ->
hex of its state; SRS / preprocessed data / proof -> dicts of the above with the reference's keys."""
from .field import FQ, FQ2, FR
from .plonk.polynomial import Polynomial
from .plonk.preprocessor import PreprocessedData
from .plonk.prover import Proof
from .plonk.srs import SRS
from .plonk.transcript import Transcript

_POLYS = ("q_l", "q_r", "q_o", "q_m", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")
_PROOF_POINTS = ("a_comm", "b_comm", "c_comm", "z_comm", "t_lo_comm", "t_mid_comm", "t_hi_comm", "W_zeta_comm", "W_zeta_omega_comm")
_PROOF_SCALARS = ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval", "r_eval")


def serialize_fr(val):
    return str(int(val))


def deserialize_fr(s):
    return FR(int(s))


def serialize_g1(point):
    return None if point is None else [str(int(point[0])), str(int(point[1]))]


def deserialize_g1(data):
    return None if data is None else (FQ(int(data[0])), FQ(int(data[1])))


def serialize_g2(point):
    if point is None:
        return None
    return [[str(int(c)) for c in point[0].coeffs], [str(int(c)) for c in point[1].coeffs]]


def deserialize_g2(data):
    if data is None:
        return None
    return (FQ2((int(data[0][0]), int(data[0][1]))), FQ2((int(data[1][0]), int(data[1][1]))))


def serialize_poly(poly):
    return None if poly is None else [str(int(c)) for c in poly.coeffs]


def deserialize_poly(data):
    return None if data is None else Polynomial([FR(int(s)) for s in data])


def serialize_fr_list(lst):
    return [str(int(v)) for v in lst]


def deserialize_fr_list(data):
    return [FR(int(s)) for s in data]


def serialize_transcript(transcript):
    return bytes(transcript.state).hex()


def deserialize_transcript(hex_str):
    t = Transcript.__new__(Transcript)
    t.state = bytearray(bytes.fromhex(hex_str))
    return t


def serialize_srs(srs):
    return {"g1_powers": [serialize_g1(p) for p in srs.g1_powers], "g2_powers": [serialize_g2(p) for p in srs.g2_powers],
            "max_degree": srs.max_degree}


def deserialize_srs(data):
    return SRS([deserialize_g1(p) for p in data["g1_powers"]], [deserialize_g2(p) for p in data["g2_powers"]], data["max_degree"])


def serialize_preprocessed(pp):
    out = {"n": pp.n, "omega": serialize_fr(pp.omega), "domain": serialize_fr_list(pp.domain), "sigma": list(pp.sigma),
           "num_public_inputs": pp.num_public_inputs}
    for name in _POLYS:
        out[name + "_poly"] = serialize_poly(getattr(pp, name + "_poly"))
        out[name + "_comm"] = serialize_g1(getattr(pp, name + "_comm"))
    return out


def deserialize_preprocessed(data):
    pp = PreprocessedData()
    pp.n = data["n"]
    pp.omega = deserialize_fr(data["omega"])
    pp.domain = deserialize_fr_list(data["domain"])
    pp.sigma = list(data["sigma"])
    pp.num_public_inputs = data["num_public_inputs"]
    for name in _POLYS:
        setattr(pp, name + "_poly", deserialize_poly(data[name + "_poly"]))
        setattr(pp, name + "_comm", deserialize_g1(data[name + "_comm"]))
    return pp


def serialize_proof(proof):
    out = {name: serialize_g1(getattr(proof, name)) for name in _PROOF_POINTS}
    for name in _PROOF_SCALARS:
        v = getattr(proof, name)
        out[name] = None if v is None else serialize_fr(v)
    return out


def deserialize_proof(data):
    proof = Proof()
    for name in _PROOF_POINTS:
        setattr(proof, name, deserialize_g1(data.get(name)))
    for name in _PROOF_SCALARS:
        v = data.get(name)
        setattr(proof, name, deserialize_fr(v) if v else None)   # the reference treats a missing / empty field as None
    return proof


def _shorten(s):
    return s if len(s) <= 8 else s[:4] + "..." + s[-4:]


def g1_short(point):
    """Display helper of the reference UI (plonk_serializers.py g1_short)."""
    if point is None:
        return "∞"
    return "(%s, %s)" % (_shorten(str(int(point[0]))), _shorten(str(int(point[1]))))


def g2_short(point):
    """plonk_serializers.py:269-280: only the x coordinate of a G2 point is shown."""
    if point is None:
        return "∞"
    return "(%s+%si, ...)" % (_shorten(str(int(point[0].coeffs[0]))), _shorten(str(int(point[0].coeffs[1]))))


def fr_short(val):
    """plonk_serializers.py:283-289 (scalars keep ten digits before they are shortened)."""
    if val is None:
        return "None"
    s = str(int(val))
    return s if len(s) <= 10 else s[:4] + "..." + s[-4:]


# Groth16 side (app.py:967-1001, 1264-1311): points persist as nested int lists.
def turn_g1_int(point):
    return None if point is None else [int(point[0]), int(point[1])]


def turn_g2_int(point):
    return None if point is None else [[int(c) for c in point[0].coeffs], [int(c) for c in point[1].coeffs]]


def g1_from_ints(data):
    return None if data is None else (FQ(data[0]), FQ(data[1]))


def g2_from_ints(data):
    return None if data is None else (FQ2((data[0][0], data[0][1])), FQ2((data[1][0], data[1][1])))
