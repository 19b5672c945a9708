"""Diffusion schedule tables (host logic, float64 numpy) for the HIP sampler.

Same tables as the reference's respacing, derived in this module's own terms: get_named_beta_schedule / betas_for_alpha_bar
(src/models/utils/gaussian_diffusion.py:236-279), space_timesteps (:1279-1332), the respaced betas of
MixerDiffusion.__init__ / MotionDiffusion.__init__ (:1436-1463, :1336-1352) and GaussianDiffusion.__init__ (:331-382).
The four fp32 coefficient rows the device kernels read are derived exactly as ``ddim_sample`` derives them: gather
from the float64 table, cast to fp32 (``_extract_into_tensor`` :1264-1277), then fp32 sqrt where the reference takes
``th.sqrt`` of the gathered fp32 tensor (:1949-1956).
"""
import math
import numpy as np


MAX_BETA = 0.999


def _cosine_alpha_bar(u):
    """Nichol & Dhariwal's squared-cosine cumulative signal level at u in [0, 1] (gaussian_diffusion.py:256-259)."""
    return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """float64 betas of the full (un-respaced) process.  "cosine": beta_i = min(1 - ab((i+1)/N) / ab(i/N), 0.999), evaluated on the
    N+1 grid points once (libm cos per point, as the reference's Python loop does, so the table is bit-identical; the ratio, the
    subtraction and the clamp are exact IEEE element-wise ops)."""
    n = int(num_diffusion_timesteps)
    if schedule_name == "linear":
        k = 1000 / n
        return np.linspace(k * 0.0001, k * 0.02, n, dtype=np.float64)
    if schedule_name == "cosine":
        grid = np.array([_cosine_alpha_bar(i / n) for i in range(n + 1)], dtype=np.float64)
        return np.minimum(1.0 - grid[1:] / grid[:-1], MAX_BETA)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def _ddim_stride(n, count):
    """Stride s of the "ddim<count>" strategies: the kept steps are 0, s, 2s, ... < n and there must be exactly `count` of them, i.e.
    ceil(n / s) == count.  ceil(n / s) is non-increasing in s, so the only candidate is the smallest s with ceil(n / s) <= count,
    s = ceil(n / count)."""
    if count <= 0:
        raise ValueError(f"ddim{count}: the number of sampling steps must be positive")
    s = -(-n // count)
    if not (1 <= s < n) or -(-n // s) != count:
        raise ValueError(f"ddim{count}: no integer stride keeps exactly {count} of {n} diffusion steps")
    return s


def space_timesteps(num_timesteps, section_counts):
    """Set of original timesteps a respaced sampler keeps (semantics of gaussian_diffusion.py:1279-1332).

    "ddimN" -> every s-th step from 0 (closed-form stride, see _ddim_stride).  Otherwise a comma-separated list (or a sequence) of
    per-section counts: the n steps are cut into len(counts) contiguous sections (the first n % len sections one step longer) and
    section j contributes counts[j] steps spread from its first to its last step; positions accumulate the fractional stride in
    float64 and round half-to-even, exactly like the reference's running sum."""
    n = int(num_timesteps)
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            return set(range(0, n, _ddim_stride(n, int(section_counts[4:]))))
        section_counts = [int(tok) for tok in section_counts.split(",")]
    counts = np.asarray(list(section_counts), dtype=np.int64)
    nsec = len(counts)
    lengths = np.full(nsec, n // nsec, dtype=np.int64)
    lengths[: n % nsec] += 1
    if (lengths < counts).any():
        j = int(np.argmax(lengths < counts))
        raise ValueError(f"section {j} has {int(lengths[j])} steps, {int(counts[j])} requested")
    firsts = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    kept = []
    for first, length, cnt in zip(firsts, lengths, counts):
        if cnt <= 0:
            continue
        step = 1.0 if cnt == 1 else (length - 1) / (cnt - 1)
        pos = np.concatenate([[0.0], np.cumsum(np.full(cnt - 1, step, dtype=np.float64))])   # sequential float64 sums: 0, s, s+s, ...
        kept.append(first + np.rint(pos).astype(np.int64))
    return set(int(v) for v in np.concatenate(kept)) if kept else set()


class RespacedSchedule:
    """float64 tables of the respaced process + the fp32 rows the kernels consume."""

    def __init__(self, betas, use_timesteps):
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.original_num_steps = len(betas)
        full = np.cumprod(1.0 - betas, axis=0)
        # kept steps in increasing order; the respaced beta of a kept step is 1 - ab(step) / ab(previous kept step)
        # (gaussian_diffusion.py:1449-1460), vectorised: same IEEE ops per element, so the tables are bit-identical
        keep = np.array(sorted(t for t in set(use_timesteps) if 0 <= t < len(betas)), dtype=np.int64)
        self.timestep_map = [int(t) for t in keep]
        ab_kept = full[keep]
        self.betas = 1 - ab_kept / np.concatenate([[1.0], ab_kept[:-1]])
        self.num_timesteps = int(self.betas.shape[0])
        self.alphas_cumprod = np.cumprod(1.0 - self.betas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)

    def key(self):
        """Content identity of the respaced process (what the device tables are a function of)."""
        return (self.original_num_steps, tuple(self.timestep_map), self.betas.tobytes())

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
