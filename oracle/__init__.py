"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU (fp32, plain PyTorch) restatement of the reference's eager path for the Style-Big-GAN custom-op hot path.
It is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under
style-big-gan_amd/ (the product) imports it, and the product has no CPU fallback.

Parity pin: every function here is checked against golden vectors captured from the reference itself
(tests/golden/make_golden.py imports /root/reference on CPU and writes tests/golden/*.npz; tests/test_oracle_golden.py
replays them).  The reference ships no tests or known-answer vectors of its own, so these captured
fixtures are the only pin.
"""
