#!/bin/bash
# Sanitizer runs of the HOST side of libgpx (sanitizer build of the shim), on the GPU box:
#   gpurun -- 'bash tools/asan_smoke.sh > gpurun_out/sanitizer_smoke.log 2>&1'
# Builds the instrumented library (host code instrumented, device code unchanged) and drives the smoke workload
# (factor, gradient, posterior, draw, batched sweep, node sweep) through it.  Python itself is not instrumented: the
# sanitizer runtime is preloaded.
#   ubsan  -fsanitize=undefined, -fno-sanitize-recover: runs everywhere.
#   asan   -fsanitize=address: ROCm's ASan runtime also intercepts hsa_amd_memory_pool_allocate (device-side ASan for
#          xnack+ targets); on gfx950 without xnack that interceptor fails the runtime's own first pool allocation
#          ("allocator is trying to allocate 0x400000 bytes", profiles/r02/sanitizer_smoke.log), so the ASan leg only
#          runs with HSA_XNACK=1 where the platform allows it and is reported, not fatal.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # $1 = make target / library suffix, $2 = runtime library name, $3.. = extra env
  kind=$1; rtname=$2; shift 2
  make -C gpax_amd/csrc $kind -j8 > /dev/null 2>&1 || { echo "$kind: build failed"; return 1; }
  RT=$(/opt/rocm/bin/hipcc -print-file-name=$rtname)
  env "$@" GPX_LIB=$PWD/gpax_amd/lib/libgpx_$kind.so LD_PRELOAD=$RT python - <<'PY'
import numpy as np
import __graft_entry__ as g
from gpax_amd import _lib
print("library under test:", _lib.lib_path())
g.smoke()
from bench_inputs import synthetic_problem, synthetic_theta_samples
X, y, Xn, _ = synthetic_problem(500, 2, 70, seed=1)
th = synthetic_theta_samples(5, 2, seed=2)
eps = np.random.default_rng(3).standard_normal((5, 1, 70))
node = _lib.Node([0], inflight=2)
m, s, i = node.predict_sweep(X, 1, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)
assert np.isfinite(m).all() and np.all(i == 0)
node.close()
e = _lib.Engine(0)
e.set_train(X)
lml, info, grad, alpha = e.fit_batch(1, th["k_length"], th["k_scale"], th["noise"], 1e-6, y)
assert np.all(info == 0) and np.isfinite(grad).all()
print("sanitizer smoke ok")
PY
  rc=$?
  rm -f gpax_amd/lib/libgpx_$kind.so
  echo "$kind leg: exit code $rc"
  return $rc
}
run ubsan libclang_rt.ubsan_standalone-x86_64.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
ub=$?
run asan libclang_rt.asan-x86_64.so HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1 || echo "asan leg did not run to completion on this platform (see header)"
exit $ub
