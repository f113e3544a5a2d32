#!/bin/bash
# AddressSanitizer run of the HOST side of libgpx (sanitizer build of the shim), on the GPU box:
#   gpurun -- 'bash tools/asan_smoke.sh > gpurun_out/asan_smoke.log 2>&1'
# Builds gpax_amd/lib/libgpx_asan.so (host code instrumented, device code unchanged) and drives the smoke workload
# (factor, gradient, posterior, draw, batched sweep, node sweep) through it.  Python itself is not instrumented:
# the ASan runtime is preloaded and leak detection is off (the interpreter "leaks" by design).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
make -C gpax_amd/csrc asan -j8 > /dev/null
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export GPX_LIB=$PWD/gpax_amd/lib/libgpx_asan.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
LD_PRELOAD=$RT python - <<'PY'
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
print("asan smoke ok")
PY
rm -f gpax_amd/lib/libgpx_asan.so
