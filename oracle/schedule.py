"""Diffusion schedule restatement (numpy, fp32 where the reference is fp32)."""
import numpy as np

f32 = np.float32


def diffusion_buffers(timesteps=1000, max_beta=0.02):
    """reference diffusion/diffusion.py:28-30, 49-82: float64 numpy, rounded once to fp32."""
    betas = np.linspace(1e-4, max_beta, timesteps)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    d = dict(
        betas=betas, alphas_cumprod=ac, alphas_cumprod_prev=ac_prev,
        sqrt_alphas_cumprod=np.sqrt(ac), sqrt_one_minus_alphas_cumprod=np.sqrt(1.0 - ac),
        log_one_minus_alphas_cumprod=np.log(1.0 - ac), sqrt_recip_alphas_cumprod=np.sqrt(1.0 / ac),
        sqrt_recipm1_alphas_cumprod=np.sqrt(1.0 / ac - 1), posterior_variance=pv,
        posterior_log_variance_clipped=np.log(np.maximum(pv, 1e-20)),
        posterior_mean_coef1=betas * np.sqrt(ac_prev) / (1.0 - ac),
        posterior_mean_coef2=(1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))
    return {k: v.astype(f32) for k, v in d.items()}


def torch_linspace_f32(start, end, steps):
    """torch.linspace on CPU for float32: step=(end-start)/(steps-1) in fp32; the first half
    counts up from start, the second half counts down from end, each element one fused
    multiply-add (ATen RangeFactories vectorised kernel; checked bit-exact against the golden grid)."""
    start, end = f32(start), f32(end)
    step = np.float64(f32((end - start) / f32(steps - 1)))
    i = np.arange(steps)
    up = (np.float64(start) + step * i).astype(f32)          # exact product, single rounding == fmaf
    dn = (np.float64(end) - step * (steps - 1 - i)).astype(f32)
    return np.where(i < steps // 2, up, dn).astype(f32)


class NoiseScheduleVP:
    """reference dpm_solver_pytorch.py:98-154 / uni_pc.py:77-134, schedule='discrete'."""

    def __init__(self, betas_f32):
        b = np.asarray(betas_f32, dtype=f32)
        la = np.log((f32(1.0) - b).astype(f32)).astype(f32)
        # torch CPU cumsum accumulates fp32 in double (acc_type) and rounds each prefix
        self.log_alpha_array = (f32(0.5) * np.cumsum(la.astype(np.float64)).astype(f32)).astype(f32)
        self.total_N = len(b)
        self.T = 1.0
        self.t_array = torch_linspace_f32(0.0, 1.0, self.total_N + 1)[1:]

    def log_alpha(self, t):
        """interpolate_fn (dpm_solver_pytorch.py:1253-1292) for one scalar fp32 t."""
        t = f32(t)
        xp, yp = self.t_array, self.log_alpha_array
        K = len(xp)
        j = int(np.searchsorted(xp, t, side="left"))
        if j == 0:
            s = 0
        elif j == K:
            s = K - 2
        else:
            s = j - 1
        sx, ex, sy, ey = xp[s], xp[s + 1], yp[s], yp[s + 1]
        return f32(sy + f32(f32(f32(t - sx) * f32(ey - sy)) / f32(ex - sx)))

    def alpha(self, t):
        return f32(np.exp(self.log_alpha(t)))

    def sigma(self, t):
        la = self.log_alpha(t)
        return f32(np.sqrt(f32(f32(1.0) - f32(np.exp(f32(f32(2.0) * la))))))

    def lam(self, t):
        la = self.log_alpha(t)
        ls = f32(f32(0.5) * f32(np.log(f32(f32(1.0) - f32(np.exp(f32(f32(2.0) * la)))))))
        return f32(la - ls)

    def time_steps(self, steps):
        """get_time_steps('time_uniform') (dpm_solver_pytorch.py:473-474): linspace(T, 1/N, steps+1)."""
        return torch_linspace_f32(self.T, 1.0 / self.total_N, steps + 1)

    def model_time(self, t):
        """get_model_input_time (dpm_solver_pytorch.py:271-280): (t - 1/N) * N in fp32."""
        return f32(f32(f32(t) - f32(1.0 / self.total_N)) * f32(self.total_N))
