"""Sampler restatements (numpy fp32).  `eps_fn(x[B,M,T], t_in[B]) -> eps[B,M,T]` is the
denoiser with the condition already bound (the reference concatenates cond on every call,
diffusion/diffusion.py:223-230)."""
import numpy as np

from .schedule import NoiseScheduleVP, f32


def _expm1(x):
    return f32(np.expm1(f32(x)))


def _full(B, v):
    return np.full((B,), v, dtype=f32)


def _x0(ns, eps_fn, x, t):
    """data_prediction_fn (dpm_solver_pytorch.py:433-442): (x - sigma*eps)/alpha."""
    eps = eps_fn(x, _full(x.shape[0], ns.model_time(t)))
    return ((x - ns.sigma(t) * eps) / ns.alpha(t)).astype(f32)


def dpm_solver_pp_2m(eps_fn, betas, x, steps):
    """DPM_Solver(algorithm_type='dpmsolver++').sample(order=2, 'time_uniform', 'multistep')
    (dpm_solver_pytorch.py:1171-1213; updates :547-580, :796-831)."""
    ns = NoiseScheduleVP(betas)
    ts = ns.time_steps(steps)
    x = x.astype(f32)
    m_prev = [_x0(ns, eps_fn, x, ts[0])]
    t_prev = [ts[0]]

    def first(x, s, t, m_s):
        h = f32(ns.lam(t) - ns.lam(s))
        phi = _expm1(-h)
        return (f32(ns.sigma(t) / ns.sigma(s)) * x - f32(ns.alpha(t) * phi) * m_s).astype(f32)

    def second(x, t):
        l1, l0, lt = ns.lam(t_prev[-2]), ns.lam(t_prev[-1]), ns.lam(t)
        h0 = f32(l0 - l1)
        h = f32(lt - l0)
        r0 = f32(h0 / h)
        D1 = (f32(f32(1.0) / r0) * (m_prev[-1] - m_prev[-2])).astype(f32)
        phi = _expm1(-h)
        ap = f32(ns.alpha(t) * phi)
        return (f32(ns.sigma(t) / ns.sigma(t_prev[-1])) * x - ap * m_prev[-1] - f32(f32(0.5) * ap) * D1).astype(f32)

    # step 1 (order 1)
    x = first(x, t_prev[-1], ts[1], m_prev[-1])
    t_prev.append(ts[1])
    m_prev.append(_x0(ns, eps_fn, x, ts[1]))
    for step in range(2, steps + 1):
        t = ts[step]
        order = min(2, steps + 1 - step) if steps < 10 else 2
        if order == 1:
            x = first(x, t_prev[-1], t, m_prev[-1])
        else:
            x = second(x, t)
        t_prev = [t_prev[-1], t]
        if step < steps:
            m_prev = [m_prev[-1], _x0(ns, eps_fn, x, t)]
    return x


def unipc_bh2(eps_fn, betas, x, steps):
    """UniPC(variant='bh2').sample(order=2, 'time_uniform', 'multistep') (uni_pc.py:590-658,
    update :471-588).  uni_pc's NoiseScheduleVP has no lambda clip (uni_pc.py:77-86); for the
    linear schedule the clip in dpm_solver_pytorch is a no-op, so one class serves both."""
    ns = NoiseScheduleVP(betas)
    ts = ns.time_steps(steps)
    x = x.astype(f32)
    m_prev = [_x0(ns, eps_fn, x, ts[0])]
    t_prev = [ts[0]]

    def update(x, t, order, use_corrector):
        l0, lt = ns.lam(t_prev[-1]), ns.lam(t)
        m0 = m_prev[-1]
        h = f32(lt - l0)
        D1 = None
        rks = []
        if order == 2:
            rk = f32(f32(ns.lam(t_prev[-2]) - l0) / h)
            rks.append(rk)
            D1 = ((m_prev[-2] - m0) / rk).astype(f32)
        rks.append(f32(1.0))
        hh = f32(-h)
        h_phi_1 = _expm1(hh)
        h_phi_k = f32(f32(h_phi_1 / hh) - f32(1.0))
        B_h = _expm1(hh)
        b = []
        fact = 1
        for i in range(1, order + 1):
            b.append(f32(f32(h_phi_k * f32(fact)) / B_h))
            fact *= i + 1
            h_phi_k = f32(f32(h_phi_k / hh) - f32(1.0 / fact))
        alpha_t = ns.alpha(t)
        x_t_ = (f32(ns.sigma(t) / ns.sigma(t_prev[-1])) * x - f32(alpha_t * h_phi_1) * m0).astype(f32)
        if D1 is not None:
            pred = (f32(0.5) * D1).astype(f32)
            x_t = (x_t_ - f32(alpha_t * B_h) * pred).astype(f32)
        else:
            x_t = x_t_
        model_t = None
        if use_corrector:
            model_t = _x0(ns, eps_fn, x_t, t)
            if order == 1:
                rhos = np.array([0.5], dtype=f32)
            else:
                R = np.array([[1.0, 1.0], [rks[0], 1.0]], dtype=f32)
                rhos = np.linalg.solve(R.astype(np.float64), np.array(b, dtype=np.float64)).astype(f32)
                # torch.linalg.solve runs LAPACK sgesv in fp32; restate the 2x2 LU in fp32
                rhos = _solve2_f32(R, np.array(b, dtype=f32))
            corr = (rhos[0] * D1).astype(f32) if D1 is not None else f32(0.0)
            D1_t = (model_t - m0).astype(f32)
            x_t = (x_t_ - f32(alpha_t * B_h) * (corr + rhos[-1] * D1_t).astype(f32)).astype(f32)
        return x_t, model_t

    x, mx = update(x, ts[1], 1, True)
    t_prev.append(ts[1])
    m_prev.append(mx)
    for step in range(2, steps + 1):
        t = ts[step]
        order = min(2, steps + 1 - step)
        x, mx = update(x, t, order, step != steps)
        t_prev = [t_prev[-1], t]
        if step < steps:
            m_prev = [m_prev[-1], mx]
    return x


def _solve2_f32(R, b):
    """sgesv on [[1,1],[r,1]] x = b with partial pivoting, in fp32."""
    a00, a01, a10, a11 = f32(R[0, 0]), f32(R[0, 1]), f32(R[1, 0]), f32(R[1, 1])
    b0, b1 = f32(b[0]), f32(b[1])
    if abs(a10) > abs(a00):
        a00, a01, a10, a11, b0, b1 = a10, a11, a00, a01, b1, b0
    l = f32(a10 / a00)
    u11 = f32(a11 - f32(l * a01))
    y1 = f32(b1 - f32(l * b0))
    x1 = f32(y1 / u11)
    x0 = f32(f32(b0 - f32(a01 * x1)) / a00)
    return np.array([x0, x1], dtype=f32)


def ddpm(eps_fn, bufs, x, k_step, noise):
    """p_sample loop (diffusion.py:95-121, 335-341): t = k_step-1 .. 0; noise[i] is the
    i-th per-step randn draw (drawn at every step, also t == 0)."""
    x = x.astype(f32)
    B = x.shape[0]
    for n, t in enumerate(reversed(range(k_step))):
        eps = eps_fn(x, np.full((B,), t, dtype=np.int64))
        x0 = (bufs["sqrt_recip_alphas_cumprod"][t] * x - bufs["sqrt_recipm1_alphas_cumprod"][t] * eps).astype(f32)
        x0 = np.clip(x0, -1.0, 1.0).astype(f32)
        mean = (bufs["posterior_mean_coef1"][t] * x0 + bufs["posterior_mean_coef2"][t] * x).astype(f32)
        mask = f32(0.0 if t == 0 else 1.0)
        sd = f32(np.exp(f32(f32(0.5) * bufs["posterior_log_variance_clipped"][t])))
        x = (mean + f32(mask * sd) * noise[n]).astype(f32)
    return x


def ddim(eps_fn, bufs, x, k_step, speedup):
    """p_sample_ddim loop (diffusion.py:123-131, 317-332)."""
    x = x.astype(f32)
    B = x.shape[0]
    ac = bufs["alphas_cumprod"]
    for t in reversed(range(0, k_step, speedup)):
        a_t = ac[t]
        a_p = ac[max(t - speedup, 0)]
        eps = eps_fn(x, np.full((B,), t, dtype=np.int64))
        c = f32(f32(np.sqrt(f32(f32(f32(1.0) - a_p) / a_p))) - f32(np.sqrt(f32(f32(f32(1.0) - a_t) / a_t))))
        x = (f32(np.sqrt(a_p)) * (x / f32(np.sqrt(a_t)) + c * eps).astype(f32)).astype(f32)
    return x


def plms(eps_fn, bufs, x, k_step, speedup):
    """p_sample_plms loop (diffusion.py:133-167, 300-316); like the reference, B == 1 only."""
    x = x.astype(f32)
    B = x.shape[0]
    ac = bufs["alphas_cumprod"]
    hist = []

    def x_pred(x, noise_t, t):
        a_t, a_p = ac[t], ac[max(t - speedup, 0)]
        sa, sp = f32(np.sqrt(a_t)), f32(np.sqrt(a_p))
        c1 = f32(f32(1.0) / f32(sa * f32(sa + sp)))
        c2 = f32(f32(1.0) / f32(sa * f32(f32(np.sqrt(f32(f32(f32(1.0) - a_p) * a_t))) + f32(np.sqrt(f32(f32(f32(1.0) - a_t) * a_p))))))
        return (x + f32(a_p - a_t) * (c1 * x - c2 * noise_t).astype(f32)).astype(f32)

    for t in reversed(range(0, k_step, speedup)):
        tt = np.full((B,), t, dtype=np.int64)
        e = eps_fn(x, tt)
        if len(hist) == 0:
            xp = x_pred(x, e, t)
            e2 = eps_fn(xp, np.full((B,), max(t - speedup, 0), dtype=np.int64))
            ep = ((e + e2) / f32(2)).astype(f32)
        elif len(hist) == 1:
            ep = ((f32(3) * e - hist[-1]) / f32(2)).astype(f32)
        elif len(hist) == 2:
            ep = ((f32(23) * e - f32(16) * hist[-1] + f32(5) * hist[-2]) / f32(12)).astype(f32)
        else:
            ep = ((f32(55) * e - f32(59) * hist[-1] + f32(37) * hist[-2] - f32(9) * hist[-3]) / f32(24)).astype(f32)
        x = x_pred(x, ep, t)
        hist.append(e)
        hist = hist[-4:]
    return x


def q_sample(bufs, x_start, t, noise):
    """q_sample (diffusion.py:169-171) at the scalar step t: sqrt(ac_t) * x_start + sqrt(1 - ac_t) * noise.
    The shallow-diffusion entry (diffusion.py:207-211) starts the sampler from q_sample(norm_spec(gt_spec), k_step - 1)."""
    return (bufs["sqrt_alphas_cumprod"][t] * x_start.astype(f32) + bufs["sqrt_one_minus_alphas_cumprod"][t] * noise.astype(f32)).astype(f32)


def sample(eps_fn, bufs, x_T, method, infer_speedup, k_step=1000, noise=None):
    """GaussianDiffusion.forward(infer=True) dispatch (diffusion.py:203-343), x_T injected.
    x_T: [B,M,T]; returns x_0 [B,M,T] (before the final transpose / acoustic_scale)."""
    t = k_step
    if method is not None and infer_speedup > 1:
        if method == "dpm-solver":
            return dpm_solver_pp_2m(eps_fn, bufs["betas"][:t], x_T, t // infer_speedup)
        if method == "unipc":
            return unipc_bh2(eps_fn, bufs["betas"][:t], x_T, t // infer_speedup)
        if method == "pndm":
            return plms(eps_fn, bufs, x_T, t, infer_speedup)
        if method == "ddim":
            return ddim(eps_fn, bufs, x_T, t, infer_speedup)
        raise NotImplementedError(method)
    return ddpm(eps_fn, bufs, x_T, t, noise)
