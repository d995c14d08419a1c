"""AddressSanitizer + UBSan over the HOST side of liblds: `make asan` compiles every source for the host
only and links it against csrc/host_stub_hip.cpp (host memory behind hipMalloc / hipMemcpy), so the code that runs before any kernel
-- weight packing of the UNet (703 tensors), of the vocoder (weight-norm folding, polyphase ConvTranspose packing) and of the LM, the
split-bf16 twin packer (which reads the packed weights back), the workspace planners and the argument validation -- executes under
the sanitizers on this GPU-less machine.  The run happens in a child process with the sanitizer runtime preloaded; any report aborts
it.  (GPU AddressSanitizer is not available on the pool; the kernels are covered by the parity suite.)"""
import glob
import os
import subprocess
import sys

from conftest import PKG, ROOT

DRIVER = r'''
import ctypes as C, sys
sys.path.insert(0, {pkg!r})
from lds import arch, init_weights, native
native.LIB_PATH = {lib!r}
L = native.lib()
cfg = arch.unet_config()
u = native.UNet(cfg, init_weights.init_state(arch.unet_param_shapes(cfg), 0))
nb = C.c_size_t()
for (B, T) in ((16, 512), (1, 77), (3, 2050)):
    native.check(L.lds_unet_workspace_bytes(u.h, B, T, C.byref(nb))); a = nb.value
    native.check(L.lds_sampler_workspace_bytes(u.h, B, T, C.byref(nb))); assert nb.value > a > 0
f32 = nb.value
try:
    u.set_gemm_mode("split_bf16")                  # the removed mode: refused
    raise SystemExit("split_bf16 was accepted")
except RuntimeError:
    pass
u.set_gemm_mode("split_f16")                       # packs the two-plane twins of every weight set
assert u.gemm_mode() == 2
native.check(L.lds_sampler_workspace_bytes(u.h, 3, 2050, C.byref(nb))); assert nb.value >= f32
u.set_gemm_mode("f32")
assert L.lds_unet_set_gemm_mode(u.h, 7) == -1 and L.lds_unet_workspace_bytes(u.h, 0, 5, C.byref(nb)) == -1
# ragged batches: the per-utterance lengths are validated before anything is launched
native.check(L.lds_unet_workspace_bytes(u.h, 3, 77, C.byref(nb)))
wsb = (C.c_char * nb.value)(); dummy = (C.c_float * 8)()
for lens, msg in (((77, 0, 5), "length[1]"), ((77, 78, 5), "length[1]")):
    rc = L.lds_unet_forward_ragged(u.h, dummy, dummy, dummy, (C.c_int32 * 3)(*lens), dummy, wsb, C.c_size_t(nb.value), 3, 77, None)
    assert rc == -1 and msg in L.lds_last_error().decode(), (rc, L.lds_last_error())
assert L.lds_unet_set_latency_mode(u.h, 2) == -1 and L.lds_unet_set_latency_mode(u.h, 1) == 0 and L.lds_unet_get_latency_mode(u.h) == 1
native.check(L.lds_unet_workspace_bytes(u.h, 1, 512, C.byref(nb))); lat = nb.value
assert L.lds_unet_set_latency_mode(u.h, 0) == 0
native.check(L.lds_unet_workspace_bytes(u.h, 1, 512, C.byref(nb))); assert lat - nb.value >= (16 << 20)      # the cluster split-K scratch
bad = dict(cfg, block_out_channels=(256, 100, 512, 512))
try:
    native.UNet(bad, {{}}); raise SystemExit("accepted an unsupported width")
except RuntimeError as e:
    assert "unsupported" in str(e), e
try:
    native.UNet(cfg, {{"conv_in.weight": init_weights.uniform("x", (4,), 0, -1, 1)}}); raise SystemExit("accepted missing weights")
except RuntimeError as e:
    assert "weight tensor" in str(e), e
h = arch.SYNTHETIC_VOCODER_H
g = native.Generator(h, init_weights.init_state(arch.generator_param_shapes(h), 0))
native.check(L.lds_vocoder_workspace_bytes(g.h, 16, 512, C.byref(nb))); assert nb.value > 0
c = arch.roformer_config()
lm = native.LM(c, arch.roformer_init_state(c, 0))
native.check(L.lds_lm_workspace_bytes(lm.h, 8, 64, 513, C.byref(nb))); assert nb.value > 0
del u, g, lm
print("sanitizer driver ok")
'''


def test_host_side_under_asan_ubsan():
    csrc = os.path.join(PKG, "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j", "8", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lib = os.path.join(csrc, "build_asan", "liblds_host_asan.so")
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    assert rt, "the sanitizer runtime of the ROCm clang is missing"
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, "-c", DRIVER.format(pkg=PKG, lib=lib)], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0 and "sanitizer driver ok" in p.stdout, (p.returncode, p.stdout[-1500:], p.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[-4000:]
