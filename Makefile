# Convenience targets; the driver uses __graft_entry__.build()/smoke() and bench.py directly.
all:
	python3 __graft_entry__.py

test:
	python3 -m pytest tests -x -q -m "not gpu"

test-gpu:
	python3 -m pytest tests -x -q -m gpu

# host C++ (libnbody_host.so, nbody) and the oracle under ASan + UBSan, then the CPU test suite against those builds
# (globals are not instrumented, -asan-globals=0: ASan trips over the merged string literal of an inline header function;
#  detect_leaks=0: the interpreter itself is not leak-clean)
test-sanitize:
	$(MAKE) -C cuda-nbody_amd/host sanitize
	LD_PRELOAD=$$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so) \
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	NBODY_HOST_LIB=$(CURDIR)/cuda-nbody_amd/san/libnbody_host.so NBODY_CLI=$(CURDIR)/cuda-nbody_amd/san/nbody \
	NBODY_ORACLE_LIB=$(CURDIR)/cuda-nbody_amd/san/liboracle.so \
	python3 -m pytest tests/test_host_cpp.py tests/test_oracle.py -x -q -m "not gpu"

bench:
	python3 bench.py

clean:
	$(MAKE) -C cuda-nbody_amd/csrc clean
	$(MAKE) -C cuda-nbody_amd/host clean
	$(MAKE) -C oracle clean

.PHONY: all test test-gpu test-sanitize bench clean
