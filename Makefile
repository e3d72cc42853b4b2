# Convenience targets; the driver uses __graft_entry__.build()/smoke() and bench.py directly.
all:
	python3 __graft_entry__.py

test:
	python3 -m pytest tests -x -q -m "not gpu"

test-gpu:
	python3 -m pytest tests -x -q -m gpu

bench:
	python3 bench.py

clean:
	$(MAKE) -C cuda-nbody_amd/csrc clean
	$(MAKE) -C cuda-nbody_amd/host clean
	$(MAKE) -C oracle clean

.PHONY: all test test-gpu bench clean
