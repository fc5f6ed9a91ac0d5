# Convenience targets; the contract entry points are __graft_entry__.py, tests/ and bench.py.
PY ?= python

build:            ## hipcc (gfx950, cross-compiles without a GPU) + gcc oracle
	$(PY) __graft_entry__.py

test:             ## CPU suite: oracle vs reference golden vectors, harness, C-ABI symbols, 2-rank gloo
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu:         ## on an MI355X: HIP vs oracle / reference digests
	$(PY) -m pytest tests -x -q -m gpu

bench:            ## one JSON line (128 pairs of 1080p per step, 3-level pyramidal)
	$(PY) bench.py

profiles:         ## regenerate profiles/<tag>_* on the GPU box
	bash tools/profiles_r04.sh

.PHONY: build test test-gpu bench profiles
