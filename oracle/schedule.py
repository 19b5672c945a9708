"""float64 diffusion schedule tables (oracle copy).

Follows /root/reference/src/models/utils/gaussian_diffusion.py:
  get_named_beta_schedule :236-260, betas_for_alpha_bar :262-279,
  GaussianDiffusion.__init__ :331-382, space_timesteps :1279-1332,
  MixerDiffusion.__init__ respacing :1436-1463 (same loop as MotionDiffusion :1336-1352).
"""
import math
import numpy as np


def cosine_betas(n, max_beta=0.999):
    f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - f((i + 1) / n) / f(i / n), max_beta) for i in range(n)])


def linear_betas(n):
    scale = 1000 / n
    return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)


def named_betas(name, n):
    if name == "cosine":
        return cosine_betas(n)
    if name == "linear":
        return linear_betas(n)
    raise NotImplementedError(f"unknown beta schedule: {name}")


def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == want:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start, out = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            out.append(start + round(cur))
            cur += stride
        start += size
    return set(out)


class Schedule:
    """Respaced tables: everything ``ddim_sample`` reads, float64."""

    def __init__(self, betas, use_timesteps):
        betas = np.array(betas, dtype=np.float64)
        base_ac = np.cumprod(1.0 - betas, axis=0)
        last, new_betas, self.timestep_map = 1.0, [], []
        use = set(use_timesteps)
        for i, ac in enumerate(base_ac):
            if i in use:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        self.betas = np.array(new_betas, dtype=np.float64)
        self.num_timesteps = len(self.betas)
        self.alphas_cumprod = np.cumprod(1.0 - self.betas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)


def make_schedule(name="cosine", steps=1000, strategy="ddim50"):
    return Schedule(named_betas(name, steps), space_timesteps(steps, strategy))
