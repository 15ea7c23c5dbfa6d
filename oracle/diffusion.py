"""DDPM schedule + reverse loop as the reference runs them (rows S1-S5).

Follows diffusion/gaussian_diffusion.py:20-64 (schedules), :120-199
(coefficient tables, fp64 numpy), :277-388 (p_mean_variance, START_X +
FIXED_SMALL + inpainting + denoised_fn hook, clip_denoised=False),
:496-548 (p_sample), :598-736 (p_sample_loop) and diffusion/respace.py:64-129
(SpacedDiffusion re-derives betas from the kept alphas_cumprod; identity map
when all steps are kept).
"""
import math
import numpy as np
import torch


def cosine_betas(steps, max_beta=0.999):
    """gaussian_diffusion.py:38-42,47-64."""
    def abar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    out = []
    for i in range(steps):
        out.append(min(1 - abar((i + 1) / steps) / abar(i / steps), max_beta))
    return np.array(out, dtype=np.float64)


def linear_betas(steps):
    """gaussian_diffusion.py:29-37."""
    scale = 1000 / steps
    return np.linspace(scale * 0.0001, scale * 0.02, steps, dtype=np.float64)


def respaced_betas(betas):
    """respace.py:73-87 with use_timesteps = all steps."""
    ac = np.cumprod(1.0 - betas)
    last, new = 1.0, []
    for a in ac:
        new.append(1 - a / last)
        last = a
    return np.array(new)


def tables(betas):
    """gaussian_diffusion.py:161-199.  All float64."""
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return dict(
        betas=betas, alphas_cumprod=ac, alphas_cumprod_prev=ac_prev,
        posterior_variance=post_var,
        posterior_log_variance_clipped=np.log(np.append(post_var[1], post_var[1:])),
        posterior_mean_coef1=betas * np.sqrt(ac_prev) / (1.0 - ac),
        posterior_mean_coef2=(1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    )


def make_schedule(steps=1000, name='cosine'):
    b = cosine_betas(steps) if name == 'cosine' else linear_betas(steps)
    return tables(respaced_betas(b))


def p_sample_loop(model, shape, sched, noise, step_noise, model_kwargs,
                  denoised_fn=None, dump_steps=None, n_steps=None, first_t=None):
    """model(x[B,1,C,T], t[B] int64, y=dict) -> x0.  ``noise`` is the initial
    image (NOT inpainted when given: gaussian_diffusion.py:691-692).
    ``step_noise(i, x)`` returns the N(0,1) draw used at loop index i (the
    reference draws th.randn_like(x) from the global generator, :532).
    Runs ``n_steps`` iterations from t = steps-1 downwards (default: all); ``first_t`` enters the
    schedule at that timestep instead, with ``noise`` taken as x_{first_t} (a window of the loop)."""
    steps = len(sched['betas'])
    y = model_kwargs['y']
    c1 = sched['posterior_mean_coef1']
    c2 = sched['posterior_mean_coef2']
    lv = sched['posterior_log_variance_clipped']
    img = noise
    dump = []
    top = steps - 1 if first_t is None else int(first_t)
    todo = top + 1 if n_steps is None else min(int(n_steps), top + 1)
    for it, i in enumerate(range(top, top - todo, -1)):
        t = torch.full((shape[0],), i, dtype=torch.int64)
        x0 = model(img, t, **model_kwargs)
        if 'inpainting_mask' in y and 'inpainted_motion' in y:
            m = y['inpainting_mask']
            x0 = x0 * (~m) + y['inpainted_motion'] * m
        if denoised_fn is not None:
            x0 = denoised_fn(x0, t, model_kwargs)
        f = img.dtype
        mean = torch.tensor(c1[i]).to(f) * x0 + torch.tensor(c2[i]).to(f) * img
        eps = step_noise(it, img)
        nz = 0.0 if i == 0 else 1.0
        img = mean + nz * torch.exp(0.5 * torch.tensor(lv[i]).to(f)) * eps
        if dump_steps is not None and it in dump_steps:
            dump.append(img.clone())
    return dump if dump_steps is not None else img
