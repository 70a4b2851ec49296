#!/bin/bash
# Host-side sanitizer build (no GPU): every translation unit of libwefax_hip.so compiled for the HOST only with AddressSanitizer +
# UndefinedBehaviorSanitizer, then the GPU-free part of the test suite through it -- plan construction (wfx_shard_layout_query,
# wfx_shard_dry_run, wfx_shard_wire_plan: every rank's exchange lists at the full 60-minute sizes), the transform planners, the
# shared-memory communicator with host buffers (wfx_comm_selftest, 2 / 3 / 5 ranks, a rank that dies, ranks that disagree), the
# exports / struct layout checks.  (GPU AddressSanitizer is not available on this pool; the device code is covered by the parity
# suites.)      bash tools/sanitize_host.sh [log]      -> profiles/<round>/sanitize_host.log
set -u
cd "$(dirname "$0")/.."
LOG=${1:-/tmp/sanitize_host.log}
B=wefax_amd/csrc/build/asan
mkdir -p $B
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
# (the device side is compiled as usual -- the fat binary must be there for the library to load -- the sanitizers instrument the host side only)
FLAGS="-O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -shared-libasan -I include -I wefax_amd/csrc"
OBJS=""
for s in wfx_context wfx_fft wfx_mrfft wfx_stages wfx_polyphase wfx_api wfx_comm wfx_dist wfx_shard wfx_synth wfx_png; do
  ( /opt/rocm/bin/hipcc $FLAGS -c wefax_amd/csrc/$s.hip -o $B/$s.o 2> $B/$s.err || echo "compile of $s failed" ) &
  OBJS="$OBJS $B/$s.o"
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
for o in $OBJS; do [ -f $o ] || { echo "missing $o" | tee $LOG; cat ${o%.o}.err | tail -5; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libasan -o $B/libwefax_hip_asan.so $OBJS -ldl -lpthread || { echo "link failed" | tee $LOG; exit 1; }
echo "built $B/libwefax_hip_asan.so (host only, ASan + UBSan)" | tee $LOG
export WFX_LIB=$PWD/$B/libwefax_hip_asan.so
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:verify_asan_link_order=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
timeout 1800 python -m pytest tests/test_sharded.py tests/test_comm_shm.py tests/test_host_logic.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tee -a $LOG | tail -5
grep -c "ERROR: AddressSanitizer\|runtime error:" $LOG | sed 's/^/sanitizer reports: /' | tee -a $LOG
rm -rf $B   # (91 MB of objects: not shipped to the GPU box)
