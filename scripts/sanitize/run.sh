#!/bin/bash
# Host side of libls_amd.so (host.c, dist.c, yaml.c) under AddressSanitizer + UBSan, CPU only (no GPU needed: everything that
# runs here is table building and parsing).  Usage: scripts/sanitize/run.sh [seconds per fuzzer, default 120] [parallel seeds, default 4]
#   1. the CPU tests that load the library (tests/test_host_tables.py, tests/test_yaml_loader.py) against the sanitizer build
#   2. mutation fuzz of the YAML loader, random-argument fuzz of the basis / operator constructors
#   3. LeakSanitizer over load -> clone -> destroy of every reference input (plain C driver, no Python in the process)
# Results of the round-5 run: profiles/r5_sanitizers_host_side.txt
set -u
cd "$(dirname "$0")/../.." || exit 1
SECS=${1:-120}; PAR=${2:-4}
W=${FUZZ_DIR:-/tmp/ls_amd_sanitize}; mkdir -p "$W"; export FUZZ_DIR=$W
GCCLIB=$(dirname "$(gcc -print-file-name=libasan.so)")
C=distributed-matvec_amd/csrc
make -C $C >/dev/null || exit 1
for f in host dist yaml; do gcc -O1 -g -std=gnu11 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -c $C/$f.c -o "$W/$f.o" || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$W/libls_amd_asan.so" "$W"/{host,dist,yaml}.o $C/{kernels,comm,util,orth,stage}.o -lm -ldl -lpthread -L"$GCCLIB" -lasan -lubsan || exit 1
export LS_AMD_LIB="$W/libls_amd_asan.so" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
PRE="$GCCLIB/libasan.so:$GCCLIB/libubsan.so"
echo "== CPU tests against the sanitizer build"
LD_PRELOAD=$PRE timeout 1800 python -m pytest tests/test_host_tables.py tests/test_yaml_loader.py -x -q -s -m "not gpu" 2>&1 | grep -E "passed|failed|runtime error|AddressSanitizer" | sort | uniq -c
echo "== fuzzers ($SECS s each, $PAR seeds)"
for s in $(seq 1 "$PAR"); do
  ( LD_PRELOAD=$PRE timeout $((SECS * 2 + 60)) python scripts/sanitize/fuzz_yaml.py "$s" "$SECS" > "$W/yaml_$s.log" 2>&1; echo "rc=$?" >> "$W/yaml_$s.log" ) &
  ( LD_PRELOAD=$PRE timeout $((SECS * 2 + 60)) python scripts/sanitize/fuzz_terms.py "$s" "$SECS" > "$W/terms_$s.log" 2>&1; echo "rc=$?" >> "$W/terms_$s.log" ) &
done
wait
for f in "$W"/yaml_*.log "$W"/terms_*.log; do echo "$(basename "$f"): $(tail -n 2 "$f" | tr '\n' ' ')"; done
echo "== LeakSanitizer: load -> clone -> destroy of the reference inputs"
gcc -g -fsanitize=address -Iinclude scripts/sanitize/lifecycle_leaks.c -o "$W/leaks" "$W/libls_amd_asan.so" -Wl,-rpath,"$W" || exit 1
ASAN_OPTIONS=detect_leaks=1 "$W/leaks" /root/reference/data/*.yaml > "$W/leaks.log" 2>&1; echo "rc=$? ($(grep -c 'leak of' "$W/leaks.log") leak reports, $(ls /root/reference/data/*.yaml | wc -l) inputs)"
echo "== ThreadSanitizer: 8 threads, object lifecycle (own objects; clones of ONE operator / basis)"
for f in host dist yaml; do gcc -O1 -g -std=gnu11 -fPIC -fsanitize=thread -c $C/$f.c -o "$W/t_$f.o" || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$W/libls_amd_tsan.so" "$W"/t_{host,dist,yaml}.o $C/{kernels,comm,util,orth,stage}.o -lm -ldl -lpthread -L"$GCCLIB" -ltsan || exit 1
for t in threads_own_objects threads_shared_basis; do
  gcc -g -fsanitize=thread -Iinclude scripts/sanitize/$t.c -o "$W/$t" "$W/libls_amd_tsan.so" -Wl,-rpath,"$W" -lpthread || exit 1
  TSAN_OPTIONS=halt_on_error=0 "$W/$t" /root/reference/data/heisenberg_kagome_12_symm.yaml /root/reference/data/heisenberg_chain_1*.yaml /root/reference/data/heisenberg_square_4x4.yaml > "$W/$t.log" 2>&1
  echo "$t: rc=$? ($(grep -c 'WARNING: ThreadSanitizer' "$W/$t.log") reports)"
done
