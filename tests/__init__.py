"""Parity tests: -m "not gpu" covers the oracle against the golden vectors, host logic and the exported C ABI;
-m gpu compares the HIP path with the oracle and the goldens through the C ABI on a real MI355X."""
