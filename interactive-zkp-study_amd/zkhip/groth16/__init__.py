"""Groth16 hot path (setup / proving) on the GPU backend; mirrors zkp/groth16 of the reference."""
