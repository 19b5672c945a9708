"""Diffusion schedule tables (host logic, float64 numpy) for the HIP sampler.

Mirrors the reference's respacing: get_named_beta_schedule / betas_for_alpha_bar
(src/models/utils/gaussian_diffusion.py:236-279), space_timesteps (:1279-1332), the respaced betas of
MixerDiffusion.__init__ / MotionDiffusion.__init__ (:1436-1463, :1336-1352) and GaussianDiffusion.__init__ (:331-382).
The four fp32 coefficient rows the device kernels read are derived exactly as ``ddim_sample`` derives them: gather
from the float64 table, cast to fp32 (``_extract_into_tensor`` :1264-1277), then fp32 sqrt where the reference takes
``th.sqrt`` of the gathered fp32 tensor (:1949-1956).
"""
import math
import numpy as np


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    betas = []
    for i in range(num_diffusion_timesteps):
        t1, t2 = i / num_diffusion_timesteps, (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(betas)


def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired_count = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired_count:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx, all_steps = 0, []
    for i, section_count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < section_count:
            raise ValueError(f"cannot divide section of {size} steps into {section_count}")
        frac_stride = 1 if section_count <= 1 else (size - 1) / (section_count - 1)
        cur_idx = 0.0
        for _ in range(section_count):
            all_steps.append(start_idx + round(cur_idx))
            cur_idx += frac_stride
        start_idx += size
    return set(all_steps)


class RespacedSchedule:
    """float64 tables of the respaced process + the fp32 rows the kernels consume."""

    def __init__(self, betas, use_timesteps):
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.original_num_steps = len(betas)
        base = np.cumprod(1.0 - betas, axis=0)
        use = set(use_timesteps)
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, ac in enumerate(base):
            if i in use:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        self.betas = np.array(new_betas, dtype=np.float64)
        self.num_timesteps = int(self.betas.shape[0])
        self.alphas_cumprod = np.cumprod(1.0 - self.betas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)

    def device_coefficients(self):
        """[4, S] fp32: c_recip, c_recipm1, sqrt(ab_prev), sqrt(1 - ab_prev)  (eta = 0 so sigma = 0)."""
        ab_prev = self.alphas_cumprod_prev.astype(np.float32)
        one = np.float32(1.0)
        return np.stack([self.sqrt_recip_alphas_cumprod.astype(np.float32),
                         self.sqrt_recipm1_alphas_cumprod.astype(np.float32),
                         np.sqrt(ab_prev),
                         np.sqrt(one - ab_prev - np.float32(0.0) ** 2)]).astype(np.float32)


def make_schedule(beta_scheduler="cosine", diffusion_steps=1000, sampling_strategy="ddim50"):
    return RespacedSchedule(get_named_beta_schedule(beta_scheduler, diffusion_steps),
                            space_timesteps(diffusion_steps, sampling_strategy))
