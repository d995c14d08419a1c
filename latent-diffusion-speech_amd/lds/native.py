"""ctypes binding of liblds.so (include/lds.h).  PyTorch is used only to own device memory and
the HIP stream; every arithmetic op of the hot path runs inside the library.  There is no CPU
fallback: if the shared object is missing or a tensor is not on a HIP device the call raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblds.so")
_lib = None

METHODS = {"dpm-solver": 1, "unipc": 2, "ddpm": 3, "ddim": 4, "pndm": 5}
TABLE_STRIDE = 16


class UNetCfg(C.Structure):
    _fields_ = [("out_dims", C.c_int), ("n_hidden", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int),
                ("norm_groups", C.c_int), ("n_blocks", C.c_int), ("block_out_channels", C.c_int * 8)]


class VocoderCfg(C.Structure):
    _fields_ = [("inter_channels", C.c_int), ("upsample_initial_channel", C.c_int), ("n_ups", C.c_int),
                ("upsample_rates", C.c_int * 8), ("upsample_kernel_sizes", C.c_int * 8), ("resblock", C.c_int),
                ("n_kernels", C.c_int), ("resblock_kernel_sizes", C.c_int * 4), ("n_dil", C.c_int),
                ("resblock_dilation_sizes", (C.c_int * 4) * 4)]


class LMCfg(C.Structure):
    _fields_ = [("hidden", C.c_int), ("heads", C.c_int), ("inter", C.c_int), ("enc_layers", C.c_int), ("dec_layers", C.c_int),
                ("text_vocab", C.c_int), ("type_vocab", C.c_int), ("sem_vocab", C.c_int), ("n_spk_rows", C.c_int), ("max_pos", C.c_int),
                ("eps", C.c_float), ("sem_bos", C.c_int), ("sem_eos", C.c_int), ("sem_pad", C.c_int)]


class ConvTest(C.Structure):
    _fields_ = [("x1", C.c_void_p), ("x2", C.c_void_p), ("C1", C.c_int), ("C2", C.c_int), ("Tsrc", C.c_int),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("Co", C.c_int), ("K", C.c_int), ("pad", C.c_int), ("dil", C.c_int),
                ("act_in", C.c_int), ("slope", C.c_float), ("res", C.c_void_p), ("epilogue", C.c_int), ("tile", C.c_int)]


class DConvTest(C.Structure):
    _fields_ = [("x1", C.c_void_p), ("x2", C.c_void_p), ("C1", C.c_int), ("C2", C.c_int), ("T", C.c_int),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("Co", C.c_int), ("K", C.c_int), ("stride", C.c_int),
                ("pad", C.c_int), ("ups", C.c_int), ("res", C.c_void_p), ("epilogue", C.c_int), ("plain_out", C.c_int),
                ("v_split", C.c_int), ("cfg", C.c_int)]


EXPORTS = [
    "lds_last_error", "lds_version", "lds_unet_create", "lds_unet_destroy", "lds_unet_workspace_bytes",
    "lds_unet_forward", "lds_sampler_run", "lds_sampler_workspace_bytes", "lds_embed_create", "lds_embed_destroy",
    "lds_embed_workspace_bytes", "lds_embed_forward", "lds_transpose", "lds_gather_rows", "lds_resample_frames", "lds_axpby", "lds_vocoder_create", "lds_vocoder_destroy",
    "lds_vocoder_workspace_bytes", "lds_vocoder_forward", "lds_lm_create", "lds_lm_destroy", "lds_lm_workspace_bytes", "lds_lm_encode",
    "lds_lm_generate", "lds_prof_enable", "lds_prof_summary", "lds_unet_set_gemm_mode", "lds_unet_get_gemm_mode",
    "lds_unet_set_latency_mode", "lds_unet_get_latency_mode", "lds_unet_forward_ragged", "lds_sampler_run_ragged", "lds_vocoder_forward_ragged"]
# include/lds_test.h: single-op entry points for tests/ and tools/ (not part of the drop-in boundary)
TEST_EXPORTS = [
    "lds_test_conv", "lds_test_dconv", "lds_bench_dconv", "lds_test_gn_apply", "lds_bench_gn_stream", "lds_test_gn_chain_k4p",
    "lds_test_ln_chain_k4p", "lds_test_attention_k4p", "lds_test_conv_transpose", "lds_test_voc_step", "lds_test_dconv_bf3",
    "lds_bench_dconv_bf3", "lds_test_k8b3_roundtrip", "lds_test_gn_apply_bf3", "lds_test_dconv_split", "lds_bench_dconv_split",
    "lds_test_split_roundtrip", "lds_test_gn_apply_split", "lds_debug_set_split_rule", "lds_test_attention_f16math",
    "lds_test_attention_latency", "lds_debug_set_gn_fold", "lds_debug_set_voc_pair", "lds_debug_set_touch_weights", "lds_test_voc_pair", "lds_test_gn_fold_k4p", "lds_bench_dconv_alt", "lds_debug_fill_u32", "lds_debug_trace",
    "lds_debug_trace_count", "lds_debug_trace_get", "lds_debug_unet_plan", "lds_test_gn_fold_split", "lds_test_cluster_join", "lds_test_lm_sample"]


def lib():
    """Load liblds.so (once).  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C latent-diffusion-speech_amd/csrc` "
                               "(there is no CPU fallback for the hot path)")
        L = C.CDLL(LIB_PATH)
        L.lds_last_error.restype = C.c_char_p
        for n in EXPORTS + TEST_EXPORTS:
            if n not in ("lds_last_error", "lds_unet_destroy", "lds_embed_destroy", "lds_vocoder_destroy", "lds_lm_destroy"):
                getattr(L, n).restype = C.c_int
        for n in ("lds_unet_destroy", "lds_embed_destroy", "lds_vocoder_destroy", "lds_lm_destroy"):
            getattr(L, n).restype = None
            getattr(L, n).argtypes = [C.c_void_p]
        L.lds_unet_set_gemm_mode.argtypes = [C.c_void_p, C.c_int]
        L.lds_unet_get_gemm_mode.argtypes = [C.c_void_p]
        L.lds_unet_set_latency_mode.argtypes = [C.c_void_p, C.c_int]
        L.lds_unet_get_latency_mode.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f"liblds error {rc}: {lib().lds_last_error().decode()}")


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype=None):
    """Device pointer of a contiguous tensor that must live on the GPU."""
    import torch
    if not t.is_cuda:
        raise RuntimeError("liblds needs tensors on a HIP device (no CPU fallback for the hot path)")
    if not t.is_contiguous():
        raise RuntimeError("liblds needs contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def _host_tensor_table(state):
    """state: name -> fp32 numpy/torch (CPU).  Returns ctypes arrays + keep-alive list."""
    names, ptrs, numel, keep = [], [], [], []
    for k, v in state.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        a = np.ascontiguousarray(a, dtype=np.float32)
        keep.append(a)
        names.append(k.encode())
        ptrs.append(a.ctypes.data)
        numel.append(a.size)
    n = len(names)
    return (n, (C.c_char_p * n)(*names), (C.c_void_p * n)(*ptrs), (C.c_int64 * n)(*numel), keep)


class Workspace:
    """Growable device scratch owned by torch: one buffer per (handle, HIP stream), grown on demand.  The library's handles are immutable
    after create, so calls on different streams may overlap as long as each has its own workspace (include/lds.h) -- which is what keying
    the buffer by the current stream gives (infer_tts.synthesize_ragged runs length buckets on several streams)."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes, device):
        import torch
        key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


class UNet:
    """Handle on the packed denoiser (lds_unet_*)."""

    def __init__(self, cfg, state):
        c = UNetCfg()
        c.out_dims, c.n_hidden = cfg["out_channels"], cfg["cond_channels"]
        c.n_layers, c.n_heads, c.norm_groups = cfg["layers_per_block"], cfg["heads"], cfg["groups"]
        boc = cfg["block_out_channels"]
        c.n_blocks = len(boc)
        for i, v in enumerate(boc):
            c.block_out_channels[i] = v
        n, names, ptrs, numel, keep = _host_tensor_table(state)
        self.h = C.c_void_p()
        check(lib().lds_unet_create(C.byref(c), n, names, ptrs, numel, C.byref(self.h)))
        self.M, self.H = c.out_dims, c.n_hidden
        self.ws = Workspace()

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.lds_unet_destroy(self.h)
            self.h = None

    def set_gemm_mode(self, mode):
        """0 / "f32": exact-fp32 MFMA (default); 2 / "split_f16": two fp16 terms per operand (opt-in; include/lds.h)"""
        m = {"f32": 0, "split_bf16": 1, "split_f16": 2}.get(mode, mode)      # (1 is refused by the library: removed mode)
        check(lib().lds_unet_set_gemm_mode(self.h, int(m)))

    def gemm_mode(self):
        return int(lib().lds_unet_get_gemm_mode(self.h))

    def set_latency_mode(self, on):
        """Tile / split choices from the actual batch (one or two utterances fill the chip); include/lds.h"""
        check(lib().lds_unet_set_latency_mode(self.h, 1 if on else 0))

    def latency_mode(self):
        return int(lib().lds_unet_get_latency_mode(self.h))

    @staticmethod
    def _lengths(lengths, B, T):
        """per-utterance frame counts of a ragged batch -> host int32 [B] (include/lds.h lds_sampler_run_ragged)"""
        a = np.ascontiguousarray(np.asarray(lengths.cpu() if hasattr(lengths, "cpu") else lengths).reshape(-1), dtype=np.int32)
        if a.shape != (B,) or a.min() < 1 or a.max() > T:
            raise ValueError(f"lengths must be {B} integers in 1 .. {T}")
        return a

    def workspace_tensor(self, B, T, device, sampler=False):
        """the caller-owned scratch a forward (or sampler run) of this size on the current stream will use (tests poison it)"""
        nb = C.c_size_t()
        check((lib().lds_sampler_workspace_bytes if sampler else lib().lds_unet_workspace_bytes)(self.h, B, T, C.byref(nb)))
        return self.ws.get(nb.value, device)

    def plan(self, B, T):
        """[(slot name, offset, bytes)] of the workspace of a forward of this size (include/lds_test.h lds_debug_unet_plan)"""
        buf = C.create_string_buffer(1 << 16)
        check(lib().lds_debug_unet_plan(self.h, B, T, buf, C.c_size_t(len(buf))))
        return [(a, int(b), int(c)) for a, b, c in (ln.split() for ln in buf.value.decode().splitlines())]

    def forward(self, x, cond, t, lengths=None):
        import torch
        B, M, T = x.shape
        assert M == self.M and cond.shape == (B, self.H, T) and t.shape == (B,)
        nb = C.c_size_t()
        check(lib().lds_unet_workspace_bytes(self.h, B, T, C.byref(nb)))
        ws = self.ws.get(nb.value, x.device)
        eps = torch.empty_like(x)
        if lengths is not None:
            ln = self._lengths(lengths, B, T)
            check(lib().lds_unet_forward_ragged(self.h, _dev(x, torch.float32), _dev(cond, torch.float32), _dev(t, torch.float32), C.c_void_p(ln.ctypes.data),
                                                _dev(eps), _dev(ws), C.c_size_t(ws.numel()), B, T, _stream()))
            return eps
        check(lib().lds_unet_forward(self.h, _dev(x, torch.float32), _dev(cond, torch.float32), _dev(t, torch.float32),
                                     _dev(eps), _dev(ws), C.c_size_t(ws.numel()), B, T, _stream()))
        return eps

    def sample(self, method, table, cond, x, noise=None, lengths=None):
        """Run a whole sampler loop in place on x [B,M,T]; table: float32 [n_rows, 16] (host); lengths: the utterances' own frame counts
        (ragged batch) or None."""
        import torch
        B, M, T = x.shape
        table = np.ascontiguousarray(table, dtype=np.float32)
        assert table.ndim == 2 and table.shape[1] == TABLE_STRIDE
        nb = C.c_size_t()
        check(lib().lds_sampler_workspace_bytes(self.h, B, T, C.byref(nb)))
        ws = self.ws.get(nb.value, x.device)
        nz = _dev(noise, torch.float32) if noise is not None else None
        if lengths is not None:
            ln = self._lengths(lengths, B, T)
            check(lib().lds_sampler_run_ragged(self.h, METHODS[method], table.shape[0], C.c_void_p(table.ctypes.data), _dev(cond, torch.float32),
                                               _dev(x, torch.float32), nz, C.c_void_p(ln.ctypes.data), _dev(ws), C.c_size_t(ws.numel()), B, T, _stream()))
            return x
        check(lib().lds_sampler_run(self.h, METHODS[method], table.shape[0], C.c_void_p(table.ctypes.data),
                                    _dev(cond, torch.float32), _dev(x, torch.float32), nz, _dev(ws),
                                    C.c_size_t(ws.numel()), B, T, _stream()))
        return x


class Embed:
    """unit_embed + spk_embed front end (lds_embed_*)."""

    def __init__(self, unit_w, unit_b, spk_w=None):
        uw = np.ascontiguousarray(unit_w, dtype=np.float32)
        ub = np.ascontiguousarray(unit_b, dtype=np.float32)
        sw = None if spk_w is None else np.ascontiguousarray(spk_w, dtype=np.float32)
        self.H, self.Cin = uw.shape
        self.h = C.c_void_p()
        check(lib().lds_embed_create(self.Cin, self.H, 0 if sw is None else sw.shape[0], C.c_void_p(uw.ctypes.data),
                                     C.c_void_p(ub.ctypes.data), None if sw is None else C.c_void_p(sw.ctypes.data),
                                     C.byref(self.h)))
        self.ws = Workspace()

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.lds_embed_destroy(self.h)
            self.h = None

    def forward(self, units, spk_id):
        import torch
        B, T, K = units.shape
        assert K == self.Cin
        nb = C.c_size_t()
        check(lib().lds_embed_workspace_bytes(self.h, B, T, C.byref(nb)))
        ws = self.ws.get(nb.value, units.device)
        cond = torch.empty(B, self.H, T, dtype=torch.float32, device=units.device)
        sid = None
        if spk_id is not None:
            sid = spk_id.reshape(B, -1)[:, 0].contiguous().to(torch.int64)
        check(lib().lds_embed_forward(self.h, _dev(units, torch.float32), _dev(sid) if sid is not None else None,
                                      _dev(cond), _dev(ws), C.c_size_t(ws.numel()), B, T, _stream()))
        return cond


def debug_fill(t, pattern):
    """every 32-bit word of a device tensor = pattern (include/lds_test.h: poisoned-workspace tests)"""
    n = t.numel() * t.element_size() // 4
    check(lib().lds_debug_fill_u32(_dev(t), C.c_size_t(n), C.c_uint32(pattern), _stream()))


def debug_trace(on):
    check(lib().lds_debug_trace(1 if on else 0))


def debug_trace_records():
    """[(name, bytes)] of the stages recorded since debug_trace(True) (host copies of every UNet stage's output)"""
    out = []
    for i in range(lib().lds_debug_trace_count()):
        name = C.create_string_buffer(96)
        ptr, nb = C.c_void_p(), C.c_size_t()
        check(lib().lds_debug_trace_get(i, name, C.c_size_t(len(name)), C.byref(ptr), C.byref(nb)))
        out.append((name.value.decode(), C.string_at(ptr, nb.value)))
    return out


def debug_trace_decode(name, raw, B):
    """one debug-trace record -> (stage name, plain [B, C, T] float32 array; a flat float32 array for records that are not activation tensors)"""
    parts = name.split("|")
    if len(parts) != 4:
        return name, np.frombuffer(raw, dtype=np.float32).copy()
    nm, Cc, T, mode = parts[0], int(parts[1]), int(parts[2]), int(parts[3])
    if mode == 0:      # K4P: [B][C/8][2][T+2][4], channel 8q + 2j + h
        a = np.frombuffer(raw, dtype=np.float32).reshape(B, Cc // 8, 2, T + 2, 4)
        out = np.empty((B, Cc // 8, 8, T), dtype=np.float32)
        for h in range(2):
            for j in range(4):
                out[:, :, 2 * j + h] = a[:, :, h, 1:T + 1, j]
        return nm, out.reshape(B, Cc, T)
    npl = 3 if mode == 1 else 2      # split planes [B][C/8][planes][T+2][8]: bf16 x 3 or fp16 x 2
    if mode == 1:
        a = (np.frombuffer(raw, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(B, Cc // 8, npl, T + 2, 8)
    else:
        a = np.frombuffer(raw, dtype=np.float16).astype(np.float32).reshape(B, Cc // 8, npl, T + 2, 8)
    v = a[:, :, 0].copy()
    for pl in range(1, npl):
        v = v + a[:, :, pl]
    return nm, np.ascontiguousarray(v[:, :, 1:T + 1].transpose(0, 1, 3, 2)).reshape(B, Cc, T)


def prof_enable(level=1):
    """0 = off, 1 = per kernel family / tile configuration, 2 = names also carry the operand shapes"""
    check(lib().lds_prof_enable(int(level)))


def prof_summary():
    """list of dicts {name,count,ms,flops,bytes}; synchronises the recorded events."""
    import json
    buf = C.create_string_buffer(1 << 16)
    check(lib().lds_prof_summary(buf, C.c_size_t(len(buf))))
    return json.loads(buf.value.decode())


def axpby(a, b, c0, c1):
    """c0*a + c1*b on the device."""
    import torch
    out = torch.empty_like(a)
    check(lib().lds_axpby(_dev(out), _dev(a, torch.float32), _dev(b, torch.float32), C.c_float(c0), C.c_float(c1),
                          C.c_int64(a.numel()), _stream()))
    return out


def gather_rows(table, idx):
    """table [N, C] fp32, idx int64 [...] -> table[idx] [..., C] on the device (codebook lookup)."""
    import torch
    idx = idx.contiguous()
    N, Cc = table.shape
    out = torch.empty(tuple(idx.shape) + (Cc,), dtype=torch.float32, device=table.device)
    if idx.numel():
        check(lib().lds_gather_rows(_dev(table, torch.float32), _dev(idx, torch.int64), _dev(out), idx.numel(), Cc, N, _stream()))
    return out


def resample_frames(x, n_out, step):
    """x [B, T, C] -> [B, n_out, C]: out[:, i] = x[:, min(floor(i * step), T - 1)] (nearest, fp32 index arithmetic)."""
    import torch
    B, T, Cc = x.shape
    out = torch.empty(B, n_out, Cc, dtype=torch.float32, device=x.device)
    if n_out:
        check(lib().lds_resample_frames(_dev(x, torch.float32), _dev(out), B, T, n_out, Cc, C.c_float(step), _stream()))
    return out


def transpose(x, scale=1.0):
    """[B,R,C] -> [B,C,R] / scale on the device."""
    import torch
    B, R, Cc = x.shape
    out = torch.empty(B, Cc, R, dtype=torch.float32, device=x.device)
    check(lib().lds_transpose(_dev(x, torch.float32), _dev(out), B, R, Cc, C.c_float(scale), _stream()))
    return out


class Generator:
    """HiFi-VAEGAN decoder (lds_vocoder_*)."""

    def __init__(self, h, state):
        c = VocoderCfg()
        c.inter_channels = h["inter_channels"]
        c.upsample_initial_channel = h["upsample_initial_channel"]
        c.n_ups = len(h["upsample_rates"])
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            c.upsample_rates[i], c.upsample_kernel_sizes[i] = u, k
        c.resblock = 1 if str(h["resblock"]) == "1" else 2
        c.n_kernels = len(h["resblock_kernel_sizes"])
        c.n_dil = len(h["resblock_dilation_sizes"][0])
        for j, (k, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            c.resblock_kernel_sizes[j] = k
            for m, d in enumerate(dil):
                c.resblock_dilation_sizes[j][m] = d
        n, names, ptrs, numel, keep = _host_tensor_table(state)
        self.h = C.c_void_p()
        check(lib().lds_vocoder_create(C.byref(c), n, names, ptrs, numel, C.byref(self.h)))
        self.hop = int(np.prod(h["upsample_rates"]))
        self.C = c.inter_channels
        self.ws = Workspace()

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.lds_vocoder_destroy(self.h)
            self.h = None

    def forward(self, z, lengths=None):
        """z [B,C,T] -> wav [B,1,T*hop]; lengths: the utterances' own frame counts (ragged batch, include/lds.h lds_vocoder_forward_ragged)"""
        import torch
        B, Cc, T = z.shape
        assert Cc == self.C
        nb = C.c_size_t()
        check(lib().lds_vocoder_workspace_bytes(self.h, B, T, C.byref(nb)))
        ws = self.ws.get(nb.value, z.device)
        wav = torch.empty(B, 1, T * self.hop, dtype=torch.float32, device=z.device)
        if lengths is not None:
            ln = UNet._lengths(lengths, B, T)
            check(lib().lds_vocoder_forward_ragged(self.h, _dev(z, torch.float32), C.c_void_p(ln.ctypes.data), _dev(wav), _dev(ws), C.c_size_t(ws.numel()),
                                                   B, T, _stream()))
            return wav
        check(lib().lds_vocoder_forward(self.h, _dev(z, torch.float32), _dev(wav), _dev(ws), C.c_size_t(ws.numel()), B, T,
                                        _stream()))
        return wav


class LM:
    """text2semantic RoFormer (lds_lm_*): encoder prefill + cached decode loop."""

    def __init__(self, cfg, state):
        c = LMCfg()
        c.hidden, c.heads, c.inter = cfg["hidden"], cfg["heads"], cfg["inter"]
        c.enc_layers, c.dec_layers = cfg["enc_layers"], cfg["dec_layers"]
        c.text_vocab, c.type_vocab, c.sem_vocab = cfg["text_vocab"], cfg["type_vocab"], cfg["sem_vocab"]
        c.n_spk_rows = cfg["n_spk"] + 1 if (cfg["n_spk"] is not None and cfg["n_spk"] > 1) else 0
        c.max_pos, c.eps = cfg["max_pos"], cfg["eps"]
        c.sem_bos, c.sem_eos, c.sem_pad = cfg["sem_bos"], cfg["sem_eos"], cfg["sem_pad"]
        n, names, ptrs, numel, keep = _host_tensor_table(state)
        self.h = C.c_void_p()
        check(lib().lds_lm_create(C.byref(c), n, names, ptrs, numel, C.byref(self.h)))
        self.cfg = cfg
        self.ws = Workspace()

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.lds_lm_destroy(self.h)
            self.h = None

    def _ws(self, B, L, max_length, device):
        nb = C.c_size_t()
        check(lib().lds_lm_workspace_bytes(self.h, B, L, max_length, C.byref(nb)))
        return self.ws.get(nb.value, device)

    def encode(self, phone, tone, spk_id=None, enc_len=None):
        """enc_len: int32 [B] on the device (real positions per right-padded row) or None"""
        import torch
        B, L = phone.shape
        ph, tn = phone.contiguous().to(torch.int64), tone.contiguous().to(torch.int64)
        sp = spk_id.contiguous().to(torch.int64) if spk_id is not None else None
        ws = self._ws(B, L, 2, phone.device)
        enc = torch.empty(B, L, self.cfg["hidden"], dtype=torch.float32, device=phone.device)
        check(lib().lds_lm_encode(self.h, _dev(ph, torch.int64), _dev(tn, torch.int64), _dev(sp, torch.int64) if sp is not None else None,
                                  _dev(enc_len, torch.int32) if enc_len is not None else None, _dev(enc), _dev(ws), C.c_size_t(ws.numel()), B, L, _stream()))
        return enc

    def generate(self, enc, max_length, do_sample, top_k, top_p, temperature, repetition_penalty, uniforms=None, return_logits=False, enc_len=None):
        import torch
        B, L, _ = enc.shape
        ws = self._ws(B, L, max_length, enc.device)
        tokens = torch.empty(B, max_length, dtype=torch.int64, device=enc.device)
        logits = torch.empty(max_length - 1, B, self.cfg["sem_vocab"], dtype=torch.float32, device=enc.device) if return_logits else None
        n = C.c_int()
        check(lib().lds_lm_generate(self.h, _dev(enc.contiguous(), torch.float32), _dev(enc_len, torch.int32) if enc_len is not None else None, B, L,
                                    int(max_length), 1 if do_sample else 0, int(top_k or 0),
                                    C.c_float(top_p), C.c_float(temperature), C.c_float(repetition_penalty),
                                    _dev(uniforms.contiguous(), torch.float32) if uniforms is not None else None, _dev(tokens),
                                    _dev(logits) if logits is not None else None, C.byref(n), _dev(ws), C.c_size_t(ws.numel()), _stream()))
        toks = tokens[:, : n.value].contiguous()
        return toks, (logits[: n.value - 1] if logits is not None else None)
