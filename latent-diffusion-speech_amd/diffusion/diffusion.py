"""`GaussianDiffusion` with the reference's constructor, buffers and `forward` signature
(reference diffusion/diffusion.py:45-343).  The sampling loops (DDPM p_sample, DDIM, PLMS,
DPM-Solver++ 2M, UniPC-bh2) run inside liblds: this module only draws x_T / per-step noise with
torch's generator (as the reference does), turns the noise schedule into a per-step table of fp32
scalars, and hands both to `lds_sampler_run`.  Training (`infer=False`) is out of scope."""
import math
from collections import deque

import numpy as np
import torch
from torch import nn

from lds import native

f32 = np.float32


def linear_beta_schedule(timesteps, max_beta=0.02):
    return np.linspace(1e-4, max_beta, timesteps)


def _linspace_f32(start, end, steps):
    # torch.linspace(float32): fp32 step, ascending half from `start`, descending half from `end`,
    # one fused multiply-add per element
    start, end = f32(start), f32(end)
    step = np.float64(f32((end - start) / f32(steps - 1)))
    i = np.arange(steps)
    up = (np.float64(start) + step * i).astype(f32)
    dn = (np.float64(end) - step * (steps - 1 - i)).astype(f32)
    return np.where(i < steps // 2, up, dn).astype(f32)


class _NoiseScheduleVP:
    """Discrete VP schedule of the solvers (reference dpm_solver_pytorch.py:98-154,1253-1292;
    uni_pc.py:77-134): piecewise-linear log(alpha_t) over t_n = (n+1)/N, all in fp32."""

    def __init__(self, betas):
        b = np.asarray(betas, dtype=f32)
        la = np.log((f32(1.0) - b).astype(f32)).astype(f32)
        self.log_alpha_array = (f32(0.5) * np.cumsum(la.astype(np.float64)).astype(f32)).astype(f32)
        self.N = len(b)
        self.t_array = _linspace_f32(0.0, 1.0, self.N + 1)[1:]

    def log_alpha(self, t):
        t = f32(t)
        xp, yp, K = self.t_array, self.log_alpha_array, self.N
        j = int(np.searchsorted(xp, t, side="left"))
        s = 0 if j == 0 else (K - 2 if j == K else j - 1)
        return f32(yp[s] + f32(f32(f32(t - xp[s]) * f32(yp[s + 1] - yp[s])) / f32(xp[s + 1] - xp[s])))

    def alpha(self, t):
        return f32(np.exp(self.log_alpha(t)))

    def sigma(self, t):
        return f32(np.sqrt(f32(f32(1.0) - f32(np.exp(f32(f32(2.0) * self.log_alpha(t)))))))

    def lam(self, t):
        la = self.log_alpha(t)
        return f32(la - f32(f32(0.5) * f32(np.log(f32(f32(1.0) - f32(np.exp(f32(f32(2.0) * la))))))))

    def grid(self, steps):
        return _linspace_f32(1.0, 1.0 / self.N, steps + 1)

    def t_in(self, t):
        return f32(f32(f32(t) - f32(1.0 / self.N)) * f32(self.N))


def _expm1(x):
    return f32(np.expm1(f32(x)))


def dpm_table(betas, steps):
    """DPM-Solver++(2M) rows (reference dpm_solver_pytorch.py:547-580,796-831,1171-1213):
    {t_in, sigma_i, alpha_i, order, sigma_{i+1}/sigma_i, alpha_{i+1}*phi_1, 0.5*alpha_{i+1}*phi_1, 1/r0}"""
    ns = _NoiseScheduleVP(betas)
    ts = ns.grid(steps)
    tab = np.zeros((steps, native.TABLE_STRIDE), dtype=f32)
    for i in range(steps):
        step = i + 1                      # update that lands on ts[step]
        order = 1 if step == 1 else (min(2, steps + 1 - step) if steps < 10 else 2)
        s, t = ts[i], ts[i + 1]
        h = f32(ns.lam(t) - ns.lam(s))
        ap = f32(ns.alpha(t) * _expm1(-h))
        row = [ns.t_in(s), ns.sigma(s), ns.alpha(s), order, f32(ns.sigma(t) / ns.sigma(s)), ap, f32(f32(0.5) * ap), 0.0]
        if order == 2:
            h0 = f32(ns.lam(s) - ns.lam(ts[i - 1]))
            row[7] = f32(f32(1.0) / f32(h0 / h))
        tab[i, :8] = row
    return tab


def _solve2(r, b0, b1):
    # torch.linalg.solve([[1,1],[r,1]], b) = fp32 LU with partial pivoting
    a00, a01, a10, a11 = f32(1.0), f32(1.0), f32(r), f32(1.0)
    if abs(a10) > abs(a00):
        a00, a01, a10, a11, b0, b1 = a10, a11, a00, a01, b1, b0
    l = f32(a10 / a00)
    u11 = f32(a11 - f32(l * a01))
    x1 = f32(f32(b1 - f32(l * b0)) / u11)
    return f32(f32(b0 - f32(a01 * x1)) / a00), x1


def unipc_table(betas, steps):
    """UniPC-bh2 rows (reference uni_pc.py:471-588,590-658).  Row 0 = first evaluation;
    row s>=1 = {t_in, sigma_s, alpha_s, order, sigma_s/sigma_{s-1}, alpha_s*h_phi_1, alpha_s*B_h, r_k,
    rho_p, rho_c0, rho_c1, use_corrector}."""
    ns = _NoiseScheduleVP(betas)
    ts = ns.grid(steps)
    tab = np.zeros((steps + 1, native.TABLE_STRIDE), dtype=f32)
    tab[0, :3] = [ns.t_in(ts[0]), ns.sigma(ts[0]), ns.alpha(ts[0])]
    for s in range(1, steps + 1):
        order = 1 if s == 1 else min(2, steps + 1 - s)
        t, p0 = ts[s], ts[s - 1]
        l0 = ns.lam(p0)
        h = f32(ns.lam(t) - l0)
        hh = f32(-h)
        h_phi_1 = _expm1(hh)
        B_h = _expm1(hh)
        h_phi_k = f32(f32(h_phi_1 / hh) - f32(1.0))
        b = []
        fact = 1
        for i in range(1, order + 1):
            b.append(f32(f32(h_phi_k * f32(fact)) / B_h))
            fact *= i + 1
            h_phi_k = f32(f32(h_phi_k / hh) - f32(1.0 / fact))
        at = ns.alpha(t)
        row = np.zeros(native.TABLE_STRIDE, dtype=f32)
        row[:7] = [ns.t_in(t), ns.sigma(t), at, order, f32(ns.sigma(t) / ns.sigma(p0)), f32(at * h_phi_1), f32(at * B_h)]
        row[8] = 0.5
        if order == 2:
            rk = f32(f32(ns.lam(ts[s - 2]) - l0) / h)
            row[7] = rk
            row[9], row[10] = _solve2(rk, b[0], b[1])
        else:
            row[7] = 1.0
            row[10] = 0.5
        row[11] = 0.0 if s == steps else 1.0
        tab[s] = row
    return tab


class GaussianDiffusion(nn.Module):
    def __init__(self, denoise_fn, out_dims=128, timesteps=1000, k_step=1000, max_beta=0.02, spec_min=-12, spec_max=2,
                 acoustic_scale=1.0):
        super().__init__()
        self.denoise_fn = denoise_fn
        self.out_dims = out_dims
        betas = linear_beta_schedule(timesteps, max_beta=max_beta)
        alphas = 1.0 - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.k_step = k_step
        self.noise_list = deque(maxlen=4)

        def reg(name, v):
            self.register_buffer(name, torch.tensor(v, dtype=torch.float32))

        # same float64 -> float32 rounding as reference diffusion.py:62-82
        reg("betas", betas)
        reg("alphas_cumprod", alphas_cumprod)
        reg("alphas_cumprod_prev", alphas_cumprod_prev)
        reg("sqrt_alphas_cumprod", np.sqrt(alphas_cumprod))
        reg("sqrt_one_minus_alphas_cumprod", np.sqrt(1.0 - alphas_cumprod))
        reg("log_one_minus_alphas_cumprod", np.log(1.0 - alphas_cumprod))
        reg("sqrt_recip_alphas_cumprod", np.sqrt(1.0 / alphas_cumprod))
        reg("sqrt_recipm1_alphas_cumprod", np.sqrt(1.0 / alphas_cumprod - 1))
        posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
        reg("posterior_variance", posterior_variance)
        reg("posterior_log_variance_clipped", np.log(np.maximum(posterior_variance, 1e-20)))
        reg("posterior_mean_coef1", betas * np.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod))
        reg("posterior_mean_coef2", (1.0 - alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - alphas_cumprod))
        self.register_buffer("spec_min", torch.FloatTensor([spec_min])[None, None, :out_dims])
        self.register_buffer("spec_max", torch.FloatTensor([spec_max])[None, None, :out_dims])
        self.acoustic_scale = acoustic_scale
        # reference diffusion.py:86-87: these instance lambdas shadow the methods of the same name
        self.norm_spec = lambda x: x * acoustic_scale
        self.denorm_spec = lambda x: x / acoustic_scale

    # ---- per-method coefficient tables (host, fp32) ----
    # Host copies of the schedule buffers and the tables built from them are cached (a device->host read per call would
    # synchronise the stream); anything that can change the buffers drops the cache.
    def _buf(self, name):
        host = self.__dict__.setdefault("_host_cache", {})
        if name not in host:
            host[name] = self.__getattr__(name).detach().cpu().numpy().astype(f32)
        return host[name]

    def _table(self, key, build):
        tabs = self.__dict__.setdefault("_table_cache", {})
        if key not in tabs:
            tabs[key] = build()
        return tabs[key]

    def _drop_host_cache(self):
        self.__dict__.pop("_host_cache", None)
        self.__dict__.pop("_table_cache", None)

    def _apply(self, fn, *args, **kwargs):
        self._drop_host_cache()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._drop_host_cache()
        return super()._load_from_state_dict(*args, **kwargs)

    def _ddpm_table(self, t):
        rc, rm1 = self._buf("sqrt_recip_alphas_cumprod"), self._buf("sqrt_recipm1_alphas_cumprod")
        c1, c2, lv = self._buf("posterior_mean_coef1"), self._buf("posterior_mean_coef2"), self._buf("posterior_log_variance_clipped")
        tab = np.zeros((t, native.TABLE_STRIDE), dtype=f32)
        for n, i in enumerate(reversed(range(t))):
            sd = f32(np.exp(f32(f32(0.5) * lv[i])))
            tab[n, :6] = [i, rc[i], rm1[i], c1[i], c2[i], f32((0.0 if i == 0 else 1.0) * sd)]
        return tab

    def _ddim_table(self, t, speedup):
        ac = self._buf("alphas_cumprod")
        rows = []
        for i in reversed(range(0, t, speedup)):
            a_t, a_p = ac[i], ac[max(i - speedup, 0)]
            c = f32(f32(np.sqrt(f32(f32(f32(1.0) - a_p) / a_p))) - f32(np.sqrt(f32(f32(f32(1.0) - a_t) / a_t))))
            rows.append([i, f32(np.sqrt(a_p)), f32(np.sqrt(a_t)), c])
        tab = np.zeros((len(rows), native.TABLE_STRIDE), dtype=f32)
        tab[:, :4] = np.array(rows, dtype=f32)
        return tab

    def _plms_table(self, t, speedup):
        ac = self._buf("alphas_cumprod")
        rows = []
        for i in reversed(range(0, t, speedup)):
            a_t, a_p = ac[i], ac[max(i - speedup, 0)]
            sa, sp = f32(np.sqrt(a_t)), f32(np.sqrt(a_p))
            c1 = f32(f32(1.0) / f32(sa * f32(sa + sp)))
            c2 = f32(f32(1.0) / f32(sa * f32(f32(np.sqrt(f32(f32(f32(1.0) - a_p) * a_t))) + f32(np.sqrt(f32(f32(f32(1.0) - a_t) * a_p))))))
            rows.append([i, max(i - speedup, 0), f32(a_p - a_t), c1, c2])
        tab = np.zeros((len(rows), native.TABLE_STRIDE), dtype=f32)
        tab[:, :5] = np.array(rows, dtype=f32)
        return tab

    def q_sample(self, x_start, t, noise=None):
        """reference diffusion.py:169-171 (scalar t; used by the shallow-diffusion entry)"""
        if noise is None:
            noise = torch.randn_like(x_start)
        ti = int(t.reshape(-1)[0]) if torch.is_tensor(t) else int(t)
        return native.axpby(x_start.contiguous(), noise.contiguous(), float(self._buf("sqrt_alphas_cumprod")[ti]),
                            float(self._buf("sqrt_one_minus_alphas_cumprod")[ti]))

    def forward(self, condition, gt_spec=None, infer=True, infer_speedup=10, method="dpm-solver", k_step=None, use_tqdm=False, *, x_T=None):
        """reference diffusion.py:189 plus one optional keyword: x_T [B,1,M,T] = the start noise to use instead of drawing it (tests, seeded
        runs; the reference draws torch.randn itself, diffusion.py:201).  The samplers' other draws (DDPM: one per step) stay torch.randn."""
        return self._sample(condition, gt_spec, infer, infer_speedup, method, k_step, None, x_T)

    def forward_ragged(self, condition, lengths, gt_spec=None, infer_speedup=10, method="dpm-solver", k_step=None, x_T=None):
        """Extension (not in the reference): a RAGGED batch in one call.  condition [B, T, H] padded to the longest utterance, lengths [B] the
        utterances' own frame counts; every utterance is sampled as if it ran alone at its own length (include/lds.h lds_sampler_run_ragged);
        frames of the result beyond an utterance's length are zeros."""
        return self._sample(condition, gt_spec, True, infer_speedup, method, k_step, lengths, x_T)

    def _sample(self, condition, gt_spec, infer, infer_speedup, method, k_step, lengths, x_T=None):
        if not infer:
            raise NotImplementedError("training (p_losses) is out of scope for the MI355X sampler build")
        if not condition.is_cuda:
            raise RuntimeError("GaussianDiffusion.forward needs tensors on a HIP device (no CPU fallback)")
        b, device = condition.shape[0], condition.device
        cond = native.transpose(condition.contiguous().float())                 # [B,H,T]
        shape = (cond.shape[0], 1, self.out_dims, cond.shape[2])
        if gt_spec is None or k_step is None:
            t = self.k_step
            if x_T is None:
                x = torch.randn(shape, device=device)
            else:
                if tuple(x_T.shape) != shape or x_T.device != device:
                    raise ValueError(f"x_T must be a {shape} tensor on {device}, got {tuple(x_T.shape)} on {x_T.device}")
                x = x_T.float().clone()      # (the sampler updates its state in place)
        else:
            t = k_step
            if x_T is not None:
                raise ValueError("x_T applies to a full run; the shallow entry starts from q_sample(gt_spec)")
            norm_spec = native.transpose(self.norm_spec(gt_spec).contiguous().float())[:, None, :, :]
            x = self.q_sample(x_start=norm_spec, t=torch.tensor([t - 1], device=device).long())
        x = x.reshape(b, self.out_dims, -1).contiguous()
        unet = self.denoise_fn.native()
        if method is not None and infer_speedup > 1:
            if method in ("dpm-solver", "unipc"):
                # multistep order 2: the reference solvers assert this (dpm_solver_pytorch.py:1172, uni_pc.py:607)
                assert t // infer_speedup >= 2, f"steps = {t} // {infer_speedup} must be >= order 2"
            if method == "dpm-solver":
                unet.sample("dpm-solver", self._table(("dpm", t, infer_speedup), lambda: dpm_table(self._buf("betas")[:t], t // infer_speedup)), cond, x, lengths=lengths)
            elif method == "unipc":
                unet.sample("unipc", self._table(("unipc", t, infer_speedup), lambda: unipc_table(self._buf("betas")[:t], t // infer_speedup)), cond, x, lengths=lengths)
            elif method == "pndm":
                if b != 1:
                    # reference diffusion.py:155 `max(t - interval, 0)` on a batch tensor raises for B > 1
                    raise RuntimeError("Boolean value of Tensor with more than one value is ambiguous")
                unet.sample("pndm", self._table(("plms", t, infer_speedup), lambda: self._plms_table(t, infer_speedup)), cond, x, lengths=lengths)
            elif method == "ddim":
                unet.sample("ddim", self._table(("ddim", t, infer_speedup), lambda: self._ddim_table(t, infer_speedup)), cond, x, lengths=lengths)
            else:
                raise NotImplementedError(method)
        else:
            tab = self._table(("ddpm", t), lambda: self._ddpm_table(t))
            chunk = 64
            for s0 in range(0, t, chunk):
                n = min(chunk, t - s0)
                # one randn per step in the reference's draw order (diffusion.py:118)
                noise = torch.stack([torch.randn(shape, device=device) for _ in range(n)]).reshape(n, b, self.out_dims, -1)
                unet.sample("ddpm", tab[s0:s0 + n], cond, x, noise.contiguous(), lengths=lengths)
        # x.squeeze(1).transpose(1, 2) / acoustic_scale  (reference diffusion.py:342-343)
        mel = native.transpose(x, float(self.acoustic_scale))
        if lengths is not None:      # the sampler's state beyond an utterance's length is the scaled start noise: not part of the result
            ln = torch.as_tensor(lengths, device=mel.device).reshape(-1, 1)
            mel = mel * (torch.arange(mel.shape[1], device=mel.device)[None, :] < ln)[:, :, None]
        return mel
