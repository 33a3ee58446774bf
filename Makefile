# Convenience targets; the build logic itself lives in vk_merkle_roots_amd/build.py (hipcc + g++, in-tree outputs).
PY ?= python

build:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test: build
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build            # on an MI355X
	$(PY) -m pytest tests -q -m gpu

smoke: build               # on an MI355X
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench: build               # on an MI355X
	$(PY) bench.py

golden:                    # needs /root/reference (regenerates tests/golden/vectors.json from the compiled reference)
	$(PY) tests/golden/make_golden.py

clean:
	rm -rf vk_merkle_roots_amd/*.so vk_merkle_roots_amd/bin oracle/*.so oracle/_ref build __pycache__ */__pycache__

.PHONY: build test test-gpu smoke bench golden clean
