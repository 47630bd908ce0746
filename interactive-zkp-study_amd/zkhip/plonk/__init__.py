"""PLONK hot path (fft / ifft / commit / SRS) on the GPU backend; mirrors zkp/plonk of the reference."""
