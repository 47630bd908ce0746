"""SHA-256 Fiat-Shamir transcript (mirrors zkp/plonk/transcript.py:36-123): state = b"plonk" ||
label || data ...; scalars are 32-byte big-endian, G1 points x || y (64 zero bytes for infinity);
a challenge hashes the state plus its label, reduces mod r and appends the digest to the state."""
import hashlib

from ..field import FR, CURVE_ORDER


class Transcript:
    def __init__(self, label=b"plonk"):
        self.state = bytearray(label)

    def append_scalar(self, label, scalar):
        self.state += label + (int(scalar) % CURVE_ORDER).to_bytes(32, "big")

    def append_point(self, label, point):
        self.state += label
        if point is None:
            self.state += bytes(64)
        else:
            self.state += int(point[0]).to_bytes(32, "big") + int(point[1]).to_bytes(32, "big")

    def challenge_scalar(self, label):
        self.state += label
        digest = hashlib.sha256(bytes(self.state)).digest()
        self.state += digest
        return FR(int.from_bytes(digest, "big"))
