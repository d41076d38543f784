# Convenience targets (the library itself is built by ragmeup_amd/build.py; __graft_entry__.build() calls that).
PY ?= python
ASAN_RT := $(firstword $(wildcard /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so))

.PHONY: all asan asan-test test clean
all:
	$(PY) -m ragmeup_amd.build

# librmu_asan.so: the library's HOST code (C-ABI, WordPiece tokenizer + thread pool, index locking / bookkeeping) under AddressSanitizer + UBSan
asan:
	$(PY) -m ragmeup_amd.build --asan

# the CPU tests that call into the library, against the sanitized build (RMU_LIB is honoured with RMU_TUNING=1; leak checking off: the
# interpreter itself is not leak-clean)
asan-test: asan
	RMU_TUNING=1 RMU_LIB=$(CURDIR)/ragmeup_amd/lib/librmu_asan.so LD_PRELOAD=$(ASAN_RT) \
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	$(PY) -m pytest tests/test_tokenizer_cpu.py tests/test_abi_cpu.py -x -q -p no:cacheprovider

test:
	$(PY) -m pytest tests -x -q -m "not gpu"

clean:
	rm -rf ragmeup_amd/lib oracle/_build
