"""CPU oracle of the DDPM hot path: test infrastructure only (see ref_cpu.py)."""
