# Convenience targets; the build itself lives in gort_amd/build.py (hipcc --offload-arch=gfx950).
.PHONY: all build oracle ref test test-gpu bench golden clean

all: build oracle

build:                      ## libgort_amd.so + gort_amd/bin/gortt (cross-compiles without a GPU)
	python3 -m gort_amd.build

oracle:                     ## CPU parity oracle (test infrastructure)
	$(MAKE) -s -C oracle libgort_oracle.so

ref:                        ## the real reference, compiled in place from /root/reference into oracle/_ref/
	$(MAKE) -s -C oracle ref

golden: ref                 ## regenerate tests/golden/ from the real reference (build container only)
	python3 tools/make_golden.py

test: build oracle          ## CPU suite: oracle vs reference goldens, host logic, ABI, sanitizers, gloo sharding
	python3 -m pytest tests -q -m "not gpu"

test-gpu: build oracle      ## parity suite on an MI355X
	python3 -m pytest tests -q -m gpu

bench: build oracle
	python3 bench.py

clean:
	rm -rf gort_amd/csrc/build gort_amd/libgort_amd.so gort_amd/bin oracle/libgort_oracle.so oracle/_ref
