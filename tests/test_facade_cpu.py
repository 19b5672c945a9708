"""CPU: host logic of the reference-API mirror (mixermdm_amd.models / configs): config parsing, state_dict names and
strictness, error behaviour.  Nothing here computes the path (that needs the GPU)."""
import os
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_cfg():
    from mixermdm_amd.configs import get_config
    return get_config(os.path.join(ROOT, "configs", "models", "MixerMDM.yaml"))


def test_config_literal_eval_like_yacs():
    cfg = make_cfg()
    assert cfg.FORCE_INFLUENCE_VAL is None            # YAML string "None" -> None
    assert cfg.GENERATOR.LATENT_DIM == 512 and cfg.MIXING_MODE == 4 and cfg.CFG_WEIGHT == 3.5
    assert cfg.STRATEGY == "ddim50" and cfg.MODEL1.endswith("individual.yaml")
    with pytest.raises(AttributeError):
        cfg.CFG_WEIGHT = 1.0                          # frozen, like yacs
    with pytest.raises(AttributeError):
        cfg.NOPE


def test_facade_surface_and_state_dict_names(golden):
    from mixermdm_amd.models import MixerMDM
    m = MixerMDM(make_cfg(), config_root=ROOT)
    assert m.sampling_strategy == "ddim50" and m.cfg_mixing_weight == 3.5 and m.mixing_mode == 4
    assert m.mixing.mode == "train" and m.mixing.force_influence_val is None and m.mixing.align is True
    keys = set(m.state_dict().keys())
    assert "mixing.denoiser1.blocks.7.sa_block.attention.in_proj_weight" in keys
    assert "mixing.influence.out.weight" in keys and m.state_dict()["mixing.influence.out.weight"].shape == (23, 512)
    assert m.state_dict()["mixing.denoiser2.blocks.0.ca_block.xf_norm.emb_layers.1.weight"].shape == (2048, 1024)
    # same relative names as the reference's Mixer.state_dict() captured in the fixture (tiny dims there)
    g, _, _ = golden("mixer")
    ref = {k[len("w:mix."):].rsplit(".", 1)[0].split("blocks.")[0] for k in g if k.startswith("w:mix.")}
    mine = {k[len("mixing."):].rsplit(".", 1)[0].split("blocks.")[0] for k in keys}
    assert ref == mine
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - (171.35e6 * 2 + 22.08e6)) / n < 0.01          # SURVEY section 6 parameter count


def test_load_state_dict_strictness():
    from mixermdm_amd.models import MixerMDM
    m = MixerMDM(make_cfg(), config_root=ROOT)
    sd = {k: torch.ones_like(v) for k, v in m.state_dict().items()}
    extra = dict(sd)
    extra["model1.decoder.net_individual.out.linear.weight"] = torch.zeros(1)     # off-path keys of the reference ckpt: ignored
    extra["clip_ln.weight"] = torch.zeros(1)
    extra["mixing.sequence_pos_encoder.pe"] = torch.zeros(1)
    m.load_state_dict(extra, strict=True)
    assert float(m.state_dict()["mixing.influence.out.bias"][0]) == 1.0
    bad = dict(sd)
    bad["mixing.bogus.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        m.load_state_dict(bad)
    sd.pop("mixing.motion_embed.weight")
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(sd)
    sd["mixing.motion_embed.weight"] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(sd)


def test_no_cpu_path_and_text_encoder_boundary():
    from mixermdm_amd.models import MixerMDM
    m = MixerMDM(make_cfg(), config_root=ROOT)
    batch = {"text_individual1": ["a"], "text_individual2": ["b"], "text_interaction": ["c"], "motion_lens": torch.tensor([[32]])}
    with pytest.raises(NotImplementedError, match="upstream of the HIP path"):
        m.forward(batch)
    batch["cond"] = torch.zeros(1, 8 * 768)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.forward(batch)


def test_sampler_args_the_reference_rejects():
    from mixermdm_amd.models import MixerDiffusion
    from mixermdm_amd.schedule import get_named_beta_schedule, space_timesteps
    d = MixerDiffusion(space_timesteps(1000, "ddim50"), betas=get_named_beta_schedule("cosine", 1000))
    assert d.num_timesteps == 50 and d.timestep_map[1] == 20
    with pytest.raises(NotImplementedError):
        d.ddim_sample_loop(None, (1, 4, 524), dump_steps=[1])
    with pytest.raises(NotImplementedError):
        d.ddim_sample_loop(None, (1, 4, 524), const_noise=True)


def test_constructor_picks_up_submodel_checkpoints_when_present(tmp_path):
    """MODEL1/MODEL2.CHECKPOINT files (mixermdm.py:43-59 formats: raw in2IN state dict; Lightning-wrapped InterGen) initialise the denoisers."""
    import torch
    import yaml
    from mixermdm_amd.configs import CfgNode
    from mixermdm_amd.models import MixerMDM
    from mixermdm_amd.synthetic import denoiser_shapes
    sub = dict(NUM_LAYERS=1, NUM_HEADS=2, DROPOUT=0.1, INPUT_DIM=262, LATENT_DIM=16, FF_SIZE=32)
    g = torch.Generator().manual_seed(0)
    sd1 = {k: torch.randn(s, generator=g) for k, s in denoiser_shapes("decoder.net_individual.", 16, 32, 1).items()}
    sd1["clip_ln_individual.weight"] = torch.ones(768)                      # other keys of the sub-model's checkpoint are ignored
    sd2 = {"model." + k: torch.randn(s, generator=g) for k, s in denoiser_shapes("decoder.net.", 16, 32, 1).items()}
    torch.save(sd1, tmp_path / "ind.ckpt")
    torch.save({"state_dict": sd2}, tmp_path / "ig.ckpt")
    yaml.safe_dump(dict(NAME="in2INind", CHECKPOINT=str(tmp_path / "ind.ckpt"), **sub), open(tmp_path / "individual.yaml", "w"))
    yaml.safe_dump(dict(NAME="InterGen", CHECKPOINT=str(tmp_path / "ig.ckpt"), **sub), open(tmp_path / "intergen.yaml", "w"))
    cfg = CfgNode(dict(NAME="MixerMDM", GENERATOR=dict(sub), DISCRIMINATOR=dict(sub), ACTIVATION="gelu", DIFFUSION_STEPS=1000, BETA_SCHEDULER="cosine",
                       SAMPLER="uniform", MOTION_REP="global", CFG_WEIGHT=3.5, MIXING_MODE=4, FORCE_INFLUENCE_VAL="None",
                       MODEL1="individual.yaml", MODEL2="intergen.yaml"))
    m = MixerMDM(cfg, num_frames=16, config_root=str(tmp_path))
    p = dict(m.named_parameters())
    assert torch.equal(p["mixing.denoiser1.blocks.0.ffn.linear1.weight"], sd1["decoder.net_individual.blocks.0.ffn.linear1.weight"])
    assert torch.equal(p["mixing.denoiser2.out.linear.weight"], sd2["model.decoder.net.out.linear.weight"])
    assert float(p["mixing.influence.out.weight"].abs().sum()) == 0.0           # the mixer's own weights stay untouched
